"""CSV goldens made by REAL pandas: the frames the reference's two drivers build cell by cell and hand to DataFrame.to_csv
(/root/reference/analysis/SSM.py:102-176 — every combination of --pick_best / --include_cys / --centrality — and
/root/reference/analysis/custom_inference.py:64,94-111), fed with a small fixed ddG table instead of a model. The native writer
(csrc/tmpnn_csv.cpp) and the Python row writer (ssm_scan.rows_for_protein / write_csv) are compared byte for byte with these files
(tests/test_host.py), so "as pandas writes it" is pinned to pandas, not to our own second implementation.

    python tests/golden/make_csv_golden.py        -> tests/golden/csv/*.csv + csv_inputs.npz     (needs pandas; no GPU, no reference import:
                                                     analysis/SSM.py pulls in omegaconf, which this image lacks, so its frame operations
                                                     are restated here operation for operation with the line they come from)
"""
import os

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "csv")
ALPHABET = "ACDEFGHIKLMNPQRSTVWYX"          # datasets.py:13


def inputs():
    """Two small 'proteins': a numbering gap ('-'), a name whose ends are eaten by str.strip('.pdb') (a character-set strip,
    SSM.py:139), a dataset wild-type string that needs quoting, and table values that exercise repr(float): exponent forms, an
    integer-valued float, zeros, a non-finite pair."""
    rng = np.random.default_rng(7)
    prots = [dict(name="dpb1x.pdb", seq="MK-LVC", wt='MK-LVC'), dict(name="2OCJ", seq="ACDW", wt='AC,D"W')]
    tabs = []
    for p in prots:
        t = rng.normal(scale=1.5, size=(len(p["seq"]), 21)).astype(np.float32)
        tabs.append(t)
    t0, t1 = tabs
    t0[0, 0], t0[0, 1], t0[0, 2] = 0.0, 1e-5, -3.0
    t0[1, 3], t0[1, 4] = 123456792.0, 2.5e-7
    t0[3, 5], t0[3, 6] = np.nan, np.inf
    t1[0, 1] = t1[0, 0] = -4.25                 # a tie: idxmin keeps the first
    t1[2, 1] = -9.0                             # best is C unless cysteine is excluded
    neigh = [np.arange(3, 3 + len(p["seq"]), dtype=np.int32) for p in prots]
    return prots, tabs, neigh


def get_ssm_mutations(seq):                     # SSM.py:16-29
    out = []
    for pos, wt in enumerate(seq):
        if wt != "-":
            out.extend(wt + str(pos) + m for m in ALPHABET[:-1])
        else:
            out.append(None)
    return out


def retrieve_best_mutants(df_slice, allow_cys=True):        # SSM.py:32-42
    best = []
    for p in df_slice.position.unique():
        p_slice = df_slice.loc[df_slice["position"] == p].reset_index(drop=True)
        if not allow_cys:
            p_slice = p_slice.loc[p_slice["mutation"] != "C"].reset_index(drop=True)
        best.append(p_slice.iloc[pd.to_numeric(p_slice["ddG_pred"]).idxmin()]["mutation"])
    return best


def ssm_frame(prots, tabs, neigh, pick_best, include_cys, centrality, model="ThermoMPNN", dataset_name="P53"):
    df = pd.DataFrame(columns=["WT Seq", "Model", "Dataset", "ddG_pred", "position", "wildtype", "mutation", "neighbors", "best_AA"])   # :102-103
    row = 0
    for p, tab, nb in zip(prots, tabs, neigh):
        muts = get_ssm_mutations(p["seq"])
        for m in muts:
            if m is None:
                continue
            wt, pos, mut = m[0], int(m[1:-1]), m[-1]                                     # :117
            vals = [float(tab[pos, ALPHABET.index(mut)]), pos, wt, mut, p["name"].strip(".pdb")]   # :138-139 (.item() of an fp32 tensor)
            for col, val in zip(["ddG_pred", "position", "wildtype", "mutation", "pdb"], vals):
                df.loc[row, col] = val                                                    # :140-141
            if centrality:
                df.loc[row, "neighbors"] = int(nb[pos])                                   # :143-144
            df.loc[row, "Model"] = model
            df.loc[row, "Dataset"] = dataset_name
            df.loc[row, "WT Seq"] = p["wt"]                                               # :146-151
            row += 1
        stripped = p["name"].strip(".pdb")
        if pick_best:                                                                     # :153-162
            cur = df.loc[df["pdb"] == stripped]
            for pos, b in zip(cur.position.unique(), retrieve_best_mutants(cur, allow_cys=include_cys)):
                df.loc[(df["pdb"] == stripped) & (df["position"] == pos), "best_AA"] = b
            df["dupe_detector"] = df["pdb"] + df["position"].astype(str)
            df = df.drop_duplicates(subset=["dupe_detector"], keep="first")
        elif not include_cys:                                                             # :164-166
            df = df.loc[df["mutation"] != "C"]
    return df.reset_index(drop=True)                                                      # :175


def custom_inference_frame(p, tab, chain="A", model="ThermoMPNN", dataset_name="2OCJ"):
    df = pd.DataFrame(columns=["Model", "Dataset", "ddG_pred", "position", "wildtype", "mutation"])         # custom_inference.py:64
    row = 0
    for m in get_ssm_mutations(p["seq"]):
        if m is None:
            continue
        wt, pos, mut = m[0], int(m[1:-1]), m[-1]
        vals = [float(tab[pos, ALPHABET.index(mut)]), pos, wt, mut, p["name"].strip(".pdb"), chain]         # :96-98
        for col, val in zip(["ddG_pred", "position", "wildtype", "mutation", "pdb", "chain"], vals):
            df.loc[row, col] = val
        df.loc[row, "Model"] = model
        df.loc[row, "Dataset"] = dataset_name
        row += 1
    return df


def main():
    os.makedirs(OUT, exist_ok=True)
    prots, tabs, neigh = inputs()
    np.savez(os.path.join(OUT, "csv_inputs.npz"), names=np.array([p["name"] for p in prots]), seqs=np.array([p["seq"] for p in prots]),
             wts=np.array([p["wt"] for p in prots]), table=np.concatenate(tabs), offsets=np.cumsum([0] + [len(p["seq"]) for p in prots]),
             neighbors=np.concatenate(neigh), pandas_version=np.array(pd.__version__))
    for pick in (False, True):
        for cys in (False, True):
            for cen in (False, True):
                df = ssm_frame(prots, tabs, neigh, pick, cys, cen)
                df.to_csv(os.path.join(OUT, f"ssm_pick{int(pick)}_cys{int(cys)}_cen{int(cen)}.csv"))        # SSM.py:176
    custom_inference_frame(prots[1], tabs[1]).to_csv(os.path.join(OUT, "custom_inference.csv"))             # custom_inference.py:111
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
