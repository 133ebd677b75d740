#!/bin/bash
# Timing-only ablations of the per-edge kernels and the featurizer (WRONG results, by construction): every variant removes one
# ingredient so that its true cost in the pipeline shows up as a time difference in ONE gpurun call. The ablations compile only
# with -DTMPNN_DEBUG_BUILD (tmpnn_common.h refuses them otherwise): the shipped library holds one form of every kernel.
#   tools/ablate.sh build      (here, no GPU needed)      -> thermompnn_amd/libtmpnn_abl_*.so
#   tools/ablate.sh run        (on the GPU box)           -> gpurun_out/ablate.log
# Round 3 (64 x L=256, f16x2; enc_edge / enc_msg / dec_msg / featurize in ms):
#   shipped 0.295 / 0.209 / 0.208 / 0.496    no GELU 0.255 / 0.192 / 0.183    no split 0.272 / 0.198 / 0.194
#   no LayerNorm statistics 0.268 (edge)      no e-tile loads 0.275 / 0.182 / 0.182    no MFMA + fragment reads 0.176 / 0.182 / 0.176 / 0.335
#   no Gaussians + no MFMA (featurizer) 0.247
cd "$(dirname "$0")/.."
L=thermompnn_amd/libtmpnn_abl
case "$1" in
build)
  for v in "nogelu -DTM_ABL_NOGELU=1" "nosplit -DTM_ABL_NOSPLIT=1" "nomfma -DTM_ABL_NOMFMA=1" "noln -DTM_ABL_NOLN=1" "noload -DTM_ABL_NOLOAD=1"; do
    set -- $v; n=$1; shift
    python -m thermompnn_amd.build --variant abl_$n -DTMPNN_DEBUG_BUILD "$@" --only=tmpnn_edge.hip --only=tmpnn_msg.hip 2>&1 | tail -1 &
  done
  python -m thermompnn_amd.build --variant abl_fnomfma -DTMPNN_DEBUG_BUILD -DTM_ABL_NOMFMA=1 --only=tmpnn_graph.hip 2>&1 | tail -1 &
  python -m thermompnn_amd.build --variant abl_fnogauss -DTMPNN_DEBUG_BUILD -DTM_ABL_NOGAUSS=1 -DTM_ABL_NOMFMA=1 --only=tmpnn_graph.hip 2>&1 | tail -1 &
  wait ;;
run)
  mkdir -p gpurun_out
  { python tools/ab_time.py shipped
    for n in nogelu nosplit nomfma noln noload fnomfma fnogauss; do TMPNN_LIB=${L}_$n.so python tools/ab_time.py $n; done; } 2>&1 | tee gpurun_out/ablate.log ;;
*) echo "usage: $0 build|run" ;;
esac
