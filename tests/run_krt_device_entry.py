"""Helper of test_krt_device_resident_entry_matches_host_entry (own process: torch first, then the library)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
with_points = len(sys.argv) > 1 and sys.argv[1] == "1"
rb = pkg.synth.make_reloc_batch(40, 96, seed_id=8, factor_type=1)
if with_points:
    rb = pkg.synth.add_reloc_points(rb, n_pt=9)
want_cam, want_summ, want_acc, _ = pkg.api.krt_solve_batch(rb)
dev = torch.device("cuda:0")


def t(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)


d_ptr, d_ref, d_cur = t(rb.match_ptr, np.int64), t(rb.uv_ref, np.float32), t(rb.uv_cur, np.float32)
d_cref, d_ccur = t(rb.cam_ref, np.float64), t(rb.cam_init, np.float64)
d_sum = torch.zeros(rb.n_query * C.sizeof(pkg.api.LmSummary), dtype=torch.uint8, device=dev)
d_acc = torch.full((rb.n_query,), -1, dtype=torch.int32, device=dev)
kw = {}
if with_points:
    kw = dict(d_point_ptr=t(rb.point_ptr, np.int64), d_pts2d=t(rb.pts2d, np.float32), d_pts3d=t(rb.pts3d, np.float64))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    pkg.api.krt_solve_batch_device(rb.n_query, d_ptr, d_ref, d_cur, d_cref, d_ccur, d_sum, d_acc, factor_type=1,
                                   stream=st.cuda_stream, **kw)
st.synchronize()
got_cam = d_ccur.cpu().numpy()
got_acc = d_acc.cpu().numpy()
summ = (pkg.api.LmSummary * rb.n_query).from_buffer_copy(d_sum.cpu().numpy().tobytes())
assert np.array_equal(got_acc, want_acc)
assert np.array_equal(got_cam, want_cam)
assert [s.num_iterations for s in summ] == [s["num_iterations"] for s in want_summ]
assert [s.final_cost for s in summ] == [s["final_cost"] for s in want_summ]
print("device entry ok", int(got_acc.sum()), "of", rb.n_query, "accepted")
