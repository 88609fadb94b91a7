import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_system_amd import _lib, faiss_compat as faiss
lib = _lib.engine()
rng = np.random.default_rng(1384)
d, n = 384, 120_000
X = rng.standard_normal((n, d)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
idx = faiss.IndexFlat(d, 0); idx.add(X); idx.search(X[:1], 1)
dev = torch.device("cuda:0"); st = torch.cuda.Stream(dev)
Q = rng.standard_normal((8, d)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
res = []
with torch.cuda.stream(st):
    qd = torch.from_numpy(Q).to(dev)
    for fn in (lib.rag_flat_search_dev, lib.rag_flat_search_dev, lib.rag_flat_search_dev_noalloc, lib.rag_flat_search_dev_noalloc):
        for nq, k in ((8, 128),):
            Dd = torch.zeros((nq, k), dtype=torch.float32, device=dev); Id = torch.zeros((nq, k), dtype=torch.int64, device=dev)
            _lib.check(fn(idx._b._h, nq, C.c_void_p(qd.data_ptr()), k, C.c_void_p(Dd.data_ptr()), C.c_void_p(Id.data_ptr()), 0, C.c_void_p(st.cuda_stream)), lib)
            st.synchronize()
            print(idx._b.last_kernel())
            res.append((Dd.cpu().numpy(), Id.cpu().numpy()))
for i in range(4):
    for j in range(i + 1, 4):
        print(i, j, "ids equal", np.array_equal(res[i][1], res[j][1]), "scores bit-equal", np.array_equal(res[i][0].view(np.uint32), res[j][0].view(np.uint32)),
              "ndiff", int((res[i][0] != res[j][0]).sum()))
