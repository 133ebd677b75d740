"""``train_thermompnn.TransferModelPL`` of the reference (/root/reference/train_thermompnn.py:20-112), inference side only:
the drivers call ``TransferModelPL.load_from_checkpoint(path, cfg=config).model`` (analysis/thermompnn_benchmarking.py:78-84).
No Lightning import: the checkpoint's ``state_dict`` is read with the restricted loader and the ``model.`` prefix stripped."""
import _repo  # noqa: F401
from thermompnn_amd.transfer_model import TransferModel
from thermompnn_amd.weights import load_thermompnn_checkpoint


class TransferModelPL:
    def __init__(self, cfg):
        self.cfg = cfg
        self.model = TransferModel(cfg)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, cfg=None, map_location=None, allow_pickle=None, **_ignored):
        if cfg is None:
            raise TypeError("load_from_checkpoint(path, cfg=config): the reference passes its OmegaConf config here")
        self = cls(cfg)
        self.model.load_state_dict(load_thermompnn_checkpoint(checkpoint_path, allow_pickle=allow_pickle))
        return self

    def eval(self):
        self.model.eval()
        return self

    def cuda(self, device=None):
        self.model.cuda(device)
        return self

    def to(self, *args, **kwargs):
        self.model.to(*args, **kwargs)
        return self

    def __call__(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    def training_step(self, *args, **kwargs):
        raise NotImplementedError("training is outside the MI355X inference engine's scope")
