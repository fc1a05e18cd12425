import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import quantize_amd.engine as engine
from test_conv_gpu import _random_case, _run_case
rng = np.random.RandomState(1)
for shp in [(2, 128, 14, 14, 256, 1, 1, 0), (1, 128, 14, 14, 128, 1, 1, 0), (2, 128, 28, 28, 128, 1, 1, 0), (1, 256, 56, 56, 64, 1, 1, 0)]:
    case = _random_case(rng, *shp, 8, 1, 8, 1, w_pc=True, a_pc=False, zeros=False, bias=True)
    y, o32, o64 = _run_case(engine, case, via_capi=True)
    err = np.abs(y.astype(np.float64) - o64)
    err[np.isnan(err)] = 1e9
    bad = np.argwhere(err > 1e-5)
    print(shp, "bad", len(bad), "nan", int(np.isnan(y).sum()), "max", err.max())
    if len(bad):
        P = y.shape[2] * y.shape[3]
        flat = bad[:, 2] * y.shape[3] + bad[:, 3]
        print(" n:", sorted(set(bad[:, 0])), "oc:", bad[:, 1].min(), bad[:, 1].max(), "n_oc", len(set(bad[:, 1])), "px:", flat.min(), flat.max(), "n_px", len(set(flat)))
        print(" px hist by 32-tile:", np.bincount(flat // 32))
        print(" oc%32 hist:", np.bincount(bad[:, 1] % 32, minlength=32))
        b = bad[0]; print(" first", b, y[tuple(b)], o64[tuple(b)])
