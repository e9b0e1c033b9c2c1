"""Diagnostic: phase breakdown of the fused residual-block kernels from in-kernel s_memtime stamps
(separate -DCOUGH_K1_STAMPS library; shares, not absolute times).  Run on the GPU box."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cough_detector_amd import _lib, build, synth
LIB = os.path.join(ROOT, "gpurun_out", "libcough_amd_stamps.so")
NAMES = ["stage (issue+LDS writes)", "barrier wait", "conv1 k-steps", "h write + barrier", "conv2 k-steps", "epilogue"]
os.makedirs(os.path.dirname(LIB), exist_ok=True)
subprocess.run([build._hipcc(), *build.FLAGS, "-DCOUGH_K1_STAMPS", "-o", LIB] + [os.path.join(build.CSRC, s) for s in build.SOURCES], check=True, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = LIB
import cough_detector_amd as cda
lib = _lib.load()
B = 4096
model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16_approx")
model.load_state_dict(synth.random_state_dict(seed=3)); model.cuda()
x = torch.rand(B, 1, 90, 101, device="cuda")
model(x)
lib.cough_debug_set_rb_stamp_buffer.argtypes = [C.c_void_p]
stamps = torch.zeros(B * 8, dtype=torch.int64, device="cuda")
assert lib.cough_debug_set_rb_stamp_buffer(stamps.data_ptr()) == 0
model(x); torch.cuda.synchronize()
assert lib.cough_debug_set_rb_stamp_buffer(None) == 0
# both block kernels wrote the same buffer; block1 ran last with ceil(B/3) workgroups, block0 with B/2 before it
st = stamps.view(B, 8).cpu().double()
n1 = (B + 2) // 3
for name, rows in (("block1 (G=3, 8 waves)", st[:n1]), ("block0 (G=1, 4 waves) [rows not overwritten by block1]", st[n1:B])):
    d = rows[:, 1:7] - rows[:, 0:6]; total = rows[:, 6] - rows[:, 0]
    print(name, "median total", float(total.median()))
    for i, n in enumerate(NAMES):
        print(f"  {n:28s} median {float(d[:, i].median()):9.0f}  share {100 * float(d[:, i].median() / total.median()):5.1f}%")
