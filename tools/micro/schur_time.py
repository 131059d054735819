"""Host-side timing of the small dense algebra of one Krylov-Schur restart (development aid)."""
import time

import numpy as np
import scipy.linalg as sla
from scipy.linalg import lapack
from threadpoolctl import threadpool_info, threadpool_limits

rng = np.random.default_rng(0)
m = 80
H = np.triu(rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)), -1)
H[40, :40] = rng.standard_normal(40)
print([(d["internal_api"], d["num_threads"]) for d in threadpool_info()])


def bench(label):
    for _ in range(2):
        t = time.perf_counter(); T, Q = sla.schur(H, output="complex"); t1 = time.perf_counter() - t
        sel = np.zeros(m, dtype=np.int32); sel[rng.permutation(m)[:50]] = 1
        t = time.perf_counter(); out = lapack.ztrsen(sel, T, Q, job="N", wantq=1); t2 = time.perf_counter() - t
        t = time.perf_counter(); w, S = sla.eig(T); t3 = time.perf_counter() - t
        t = time.perf_counter(); r = lapack.zgees(lambda z: None, H, sort_t=0); t4 = time.perf_counter() - t
    print(f"{label}: schur {t1 * 1e3:.2f} ms  trsen(50 of 80) {t2 * 1e3:.2f} ms  eig(T) {t3 * 1e3:.2f} ms  raw zgees {t4 * 1e3:.2f} ms")


bench("default threads")
with threadpool_limits(1):
    bench("1 thread")
