"""The native CSV writer alone on BASELINE config 3's table (5.9 M rows, 2.43 GB of text to tmpfs), six runs: the run-to-run spread
of the end-to-end CSV leg is the tmpfs write itself (0.30-0.80 s warm, alternating with the release of the previous file's pages;
1.0-1.5 s for the first run of a process). Pre-sizing the file (ftruncate to an upper bound) was measured: no difference.
    python tools/csv_ab.py        (GPU box, 16 usable CPUs)"""
import numpy as np, time, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from thermompnn_amd import native_csv
rng=np.random.default_rng(0); AA="ACDEFGHIKLMNPQRSTVWY"
lens=np.random.default_rng(1).integers(64,513,size=1024)
seqs=[''.join(AA[k] for k in rng.integers(0,20,L)) for L in lens]
table=rng.normal(size=(int(lens.sum()),21)).astype(np.float32); off=np.concatenate([[0],np.cumsum(lens)]).astype(np.int32)
names=[f'p{i}' for i in range(1024)]
ts=[]
for rep in range(6):
    t=time.perf_counter(); w=native_csv.CsvWriter('/dev/shm/ab.csv'); w.write_ssm(table,off,seqs,names,n_threads=15,include_cys=True); n=w.close(); ts.append(time.perf_counter()-t); os.remove('/dev/shm/ab.csv')
print(os.environ.get('TMPNN_LIB','shipped')[-20:], ' '.join(f'{x:.3f}' for x in ts), 'rows', n)
