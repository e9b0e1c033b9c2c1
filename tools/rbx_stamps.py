"""Diagnostic: phase breakdown of the split-bf16 residual-block kernels (resblock_x3.h) from in-kernel s_memtime
stamps of wave 0 (separate -DCOUGH_K1_STAMPS library; shares, not absolute times).  Run on the GPU box."""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cough_detector_amd import _lib, build, synth
LIB = os.path.join(ROOT, "gpurun_out", "libcough_amd_stamps.so")
NAMES = ["stage: loads + split + LDS writes", "barrier after staging", "conv1 + projection k-steps", "barrier + h write + barrier",
         "conv2 k-steps", "barrier + output tile + barrier", "global store (+ head)"]
os.makedirs(os.path.dirname(LIB), exist_ok=True)
if os.environ.get("RBX_STAMPS_LIB"):      # a library prebuilt with -DCOUGH_K1_STAMPS (+ variant flags), e.g. by tools/build_variant.sh
    LIB = os.path.abspath(os.environ["RBX_STAMPS_LIB"])
    print("==", os.path.basename(LIB))
else:
    subprocess.run([build._hipcc(), *build.FLAGS, "-DCOUGH_K1_STAMPS", "-o", LIB] + [os.path.join(build.CSRC, s) for s in build.SOURCES], check=True, stderr=subprocess.DEVNULL)
_lib.LIB_PATH = LIB
import cough_detector_amd as cda
lib = _lib.load()
B = int(os.environ.get("RBX_STAMPS_B", "4096"))
model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16x3")
model.load_state_dict(synth.random_state_dict(seed=3)); model.cuda()
x = torch.rand(B, 1, 90, 101, device="cuda")
for _ in range(20):
    model(x)
lib.cough_debug_set_rb_stamp_buffer.argtypes = [C.c_void_p]
stamps = torch.zeros(B * 8, dtype=torch.int64, device="cuda")   # block 0 writes row blockIdx.x < B
assert lib.cough_debug_set_rb_stamp_buffer(stamps.data_ptr()) == 0
model(x); torch.cuda.synchronize()
assert lib.cough_debug_set_rb_stamp_buffer(None) == 0
# both block kernels wrote the same buffer: block 1 ran last with B/2 workgroups, block 0 with B before it
st = stamps.view(B, 8).cpu().double()
n1 = (B + 1) // 2
for name, rows in (("block1 (G=2)", st[:n1]), ("block0 (G=1) [rows not overwritten by block1]", st[n1:B])):
    d = rows[:, 1:8] - rows[:, 0:7]; total = rows[:, 7] - rows[:, 0]
    span = rows[:, 7].max() - rows[:, 0].min()
    print(name, "median workgroup lifetime", float(total.median()), "shader cycles; kernel span", float(span), "cycles")
    for i, n in enumerate(NAMES):
        print(f"  {n:36s} median {float(d[:, i].median()):9.0f}  share {100 * float(d[:, i].median() / total.median()):5.1f}%")
