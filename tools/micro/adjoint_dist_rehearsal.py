import sys, socket, tempfile
from pathlib import Path
import numpy as np
root = Path("/root/repo")
sys.path[:0] = [str(root), str(root / "lsa-fw_amd"), str(root / "tests")]
import torch.multiprocessing as mp
from test_gpu_sharded import _rank_adjoint
from synthetic import fem
from oracle import shift_invert

if __name__ == "__main__":
    case, world, env = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_rank_adjoint, args=(world, port, tmp, case, env), nprocs=world, join=True)
        out = [dict(np.load(Path(tmp) / f"adj{r}.npz")) for r in range(world)]
    same = all(np.array_equal(o["lam"], out[0]["lam"]) and np.array_equal(o["V"], out[0]["V"]) for o in out)
    es = fem.cube_case(case)
    AH, MH = es.A.conj().T.tocsr(), es.M.conj().T.tocsr()
    res = shift_invert.compute_residuals(AH, MH, out[0]["lam"], out[0]["V"]).max()
    print(case, world, env, "ranks identical", same, "dist nodes", int(out[0]["ndist"]), "adjoint residual %.2e" % res, "gmres", int(out[0]["gmres"]), "lam", out[0]["lam"][:2])
