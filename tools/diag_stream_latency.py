import gc, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cough_detector_amd as cda
from cough_detector_amd import synth
from cough_detector_amd.streaming import MultiStreamDetector
torch.set_num_threads(4)
S=64
model = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16_approx")
model.load_state_dict(synth.random_state_dict(seed=3))
det = MultiStreamDetector(model, S, confidence_threshold=0.7, clock=lambda: 0.0)
audio = torch.from_numpy(np.stack([synth.make_stream(100+s, 30.0) for s in range(S)])).pin_memory()
def run(tag):
    det.reset(); lat=[]
    for i in range(0, audio.shape[1]-1600+1, 1600):
        t0=time.perf_counter(); det.push(audio[:, i:i+1600]); lat.append((time.perf_counter()-t0)*1e3)
    lat=np.array(lat); big=np.nonzero(lat>5)[0]
    print(tag, "p50 %.3f p99 %.3f max %.3f"%(np.percentile(lat,50),np.percentile(lat,99),lat.max()), "big ticks", big.tolist()[:12], lat[big][:6].round(1).tolist())
run("warm")
run("default")
gc.disable(); run("gc off"); gc.enable()
# isolate: only the GPU part repeated with fixed inputs
w = torch.randn(64,16000,device="cuda")
lat=[]
for k in range(300):
    t0=time.perf_counter(); f=det.pre.featurize_batch(w, normalize=True); _,p=model.predict(f.unsqueeze(1)); p[:,1].cpu(); lat.append((time.perf_counter()-t0)*1e3)
lat=np.array(lat); print("gpu only p50 %.3f max %.3f"%(np.percentile(lat,50),lat.max()), np.nonzero(lat>5)[0].tolist()[:10])
lat=[]
for k in range(300):
    t0=time.perf_counter(); x=audio[:, :1600].to("cuda", non_blocking=True); torch.cuda.synchronize(); lat.append((time.perf_counter()-t0)*1e3)
lat=np.array(lat); print("h2d only p50 %.3f max %.3f"%(np.percentile(lat,50),lat.max()), np.nonzero(lat>5)[0].tolist()[:10])
