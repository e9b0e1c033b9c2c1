"""Numerical study (CPU, float64 reference): which operand precisions keep the residual net within the 1e-3 logit tolerance.
Result (profiles/r05_classifier_operand_precision.txt): single-f16 activations with split-f16 weights (2 MFMAs per product instead of
3) are 2e-3..8e-3 off -- outside the tolerance; split-bf16 both sides (the shipped bf16x3) is 7e-5..2e-4.  Not built."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import resnet as ores
from parity import realistic_state_dict, synth_batch
from oracle import featurizer as ofeat
torch.manual_seed(0)

def fold(w, b, sd, bn):
    s = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
    return (w.double() * s[:, None, None, None]), ((b.double() - sd[bn + ".running_mean"].double()) * s + sd[bn + ".bias"].double())

def q_act(x, mode):
    if mode == "f32": return x
    if mode == "f16": return x.float().half().double()
    if mode == "bf16x2":
        h = x.float().bfloat16().float(); l = (x.float() - h).bfloat16().float(); return (h + l).double()
    if mode == "bf16": return x.float().bfloat16().double()
def q_w(w, mode):
    if mode == "f32": return w
    if mode == "f16": return w.float().half().double()
    if mode == "f16x2":
        h = w.float().half().float(); l = (w.float() - h).half().float(); return (h + l).double()
    if mode == "bf16x2":
        h = w.float().bfloat16().float(); l = (w.float() - h).bfloat16().float(); return (h + l).double()

def fwd(x, sd, am, wm, stem_exact=True):
    x = x.double()
    w, b = fold(sd["conv1.0.weight"], sd["conv1.0.bias"], sd, "conv1.1")
    sa, sw = ("bf16x2", "bf16x2") if stem_exact else (am, wm)
    y = F.conv2d(q_act(x, sa), q_w(w, sw), b, stride=2, padding=3)
    a = F.max_pool2d(F.relu(y), 2).float().double()
    for i in range(2):
        p = f"res_blocks.{i}"
        ws, bs = fold(sd[p + ".skip.0.weight"], sd[p + ".skip.0.bias"], sd, p + ".skip.1")
        w1, b1 = fold(sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], sd, p + ".bn1")
        w2, b2 = fold(sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], sd, p + ".bn2")
        aq = q_act(a, am)
        idn = F.conv2d(aq, q_w(ws, wm), bs, stride=2)
        m = F.relu(F.conv2d(aq, q_w(w1, wm), b1, stride=2, padding=1)).float().double()
        o = F.conv2d(q_act(m, am), q_w(w2, wm), b2, padding=1)
        a = F.relu(o + idn).float().double()
    return F.linear(a.mean(dim=(2, 3)), sd["fc.2.weight"].double(), sd["fc.2.bias"].double())

for seed in (11, 13, 5):
    sd = realistic_state_dict(seed)
    w = synth_batch(100 + seed, 256, peak_normalize=False)
    x = ofeat.extract_features_batch(w, normalize_first=True)[:, None]
    ref = fwd(x, sd, "f32", "f32")
    o32 = ores.forward(x, sd).double()
    print(f"seed {seed}: f32 oracle vs f64 {float((o32-ref).abs().max()):.2e}  logits std {float(ref.std()):.2f}")
    for am, wm in [("bf16x2", "bf16x2"), ("f16", "f16x2"), ("f16", "f16"), ("bf16", "bf16x2")]:
        for se in (True, False):
            e = (fwd(x, sd, am, wm, se) - ref).abs()
            print(f"   act {am:7s} w {wm:7s} stem_exact {se}: max {float(e.max()):.2e} mean {float(e.mean()):.2e}")
