"""C5 probe: batched relocalization, N queries x 128 matches (queries/s and LM iterations/s; device time only)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()  # torch's ROCm runtime first ("next to PyTorch")
import __graft_entry__ as ge
pkg = ge.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ft = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t = time.time(); base = pkg.synth.make_reloc_batch(2000, 128, seed_id=ft, factor_type=ft); tg = time.time() - t
rep = (N + base.n_query - 1) // base.n_query
rb = pkg.synth.RelocBatch(n_query=base.n_query * rep, match_ptr=np.arange(base.n_query * rep + 1, dtype=np.int64) * 128,
                          uv_ref=np.tile(base.uv_ref, (rep, 1)), uv_cur=np.tile(base.uv_cur, (rep, 1)), cam_ref=np.tile(base.cam_ref, (rep, 1)),
                          cam_init=np.tile(base.cam_init, (rep, 1)), cam_gt=np.tile(base.cam_gt, (rep, 1)), factor_type=ft)
pkg.api.krt_solve_batch(rb)
cam, summ, acc, ms = pkg.api.krt_solve_batch(rb)
its = sum(s["num_lm_steps"] for s in summ)
print(json.dumps(dict(queries=rb.n_query, factor=ft, device_ms=ms, queries_per_s=rb.n_query / ms * 1e3, lm_it_per_s=its / ms * 1e3,
                      mean_lm_steps=its / rb.n_query, accepted=float(acc.mean()), gen_s=tg,
                      algorithmic_GBps=its * (2 * 16 * 128 + 240) / ms / 1e6)))

# the same queries resident in HBM, through ptz_krt_solve_batch_device (no PCIe, no host staging): wall time of call + sync
import ctypes as C
dev = torch.device("cuda:0")
t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
d_ptr, d_ref, d_cur = t(rb.match_ptr, np.int64), t(rb.uv_ref, np.float32), t(rb.uv_cur, np.float32)
d_cref, d_init = t(rb.cam_ref, np.float64), t(rb.cam_init, np.float64)
d_sum = torch.zeros(rb.n_query * C.sizeof(pkg.api.LmSummary), dtype=torch.uint8, device=dev)
d_acc = torch.zeros(rb.n_query, dtype=torch.int32, device=dev)
best = 1e9
for _ in range(5):
    d_ccur = d_init.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pkg.api.krt_solve_batch_device(rb.n_query, d_ptr, d_ref, d_cur, d_cref, d_ccur, d_sum, d_acc, factor_type=ft)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(json.dumps(dict(resident_entry_wall_ms=best * 1e3, resident_queries_per_s=rb.n_query / best, accepted=float(d_acc.float().mean().item()))))
