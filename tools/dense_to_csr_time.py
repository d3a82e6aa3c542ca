import sys, time, json
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
import numpy as np, torch
pkg = ge.load_package(); capi = pkg.capi; W = pkg.workloads
dev = torch.device("cuda:0")
for (M, N, z) in [(4096, 4096, 0.5), (4096, 4096, 0.99), (16384, 16384, 0.9)]:
    rng = np.random.Generator(np.random.PCG64(1))
    A = rng.uniform(-1, 1, size=(M, N)).astype(np.float32)
    A[rng.random(size=(M, N)) < z] = 0
    dA = torch.from_numpy(A).to(dev)
    torch.cuda.synchronize()
    for rep in range(3):
        t = time.perf_counter()
        h = capi.CsrMatrix.from_dense_device(dA)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        nnz = h.nnz
        h.close()
    t = time.perf_counter(); h = capi.CsrMatrix.from_dense_host(A); dth = time.perf_counter() - t; h.close()
    print(json.dumps(dict(M=M, N=N, zero=z, nnz=nnz, device_ms=round(dt * 1e3, 3), dense_GBs=round(M * N * 4 / dt / 1e9, 1), from_host_ms=round(dth * 1e3, 2))))
