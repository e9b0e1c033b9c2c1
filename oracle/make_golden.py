"""Generate the committed golden fixtures under ``tests/golden/``.

Run in the BUILD CONTAINER only (needs ``/root/reference``):

    python -m oracle.make_golden

* ``resnet_golden.npz`` -- produced by executing the REFERENCE module
  ``/root/reference/src/model.py`` (imported by file path; it needs only
  torch): seeded weights with randomised BatchNorm statistics, 32 feature
  inputs (BASELINE.json configs[0]: "32 clips"), activations after the stem /
  block 0 / block 1, logits, softmax, argmax.  The head (``fc.2``) is scaled
  and re-centred so that the logits look like a TRAINED detector's -- class
  margins with a standard deviation of 2.5 and |logit| up to ~5 -- because a
  default-init head (|w| <= 0.088, margin spread 0.014) would hide any
  reduced-precision error of the conv stack behind a near-degenerate Linear
  layer.  This is what pins ``oracle/resnet.py`` and the HIP classifier.
* ``features_golden.npz`` -- produced by ``oracle/featurizer.py`` (the
  torch-CPU restatement; torchaudio itself is unavailable, so these vectors are
  labelled "oracle-generated", not "reference-generated") for synthetic clips
  ``seed = 0..31`` plus a CRC of each waveform so the GPU box can confirm it
  regenerated identical inputs.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cough_detector_amd import synth            # noqa: E402
from oracle import featurizer                    # noqa: E402

REF_MODEL = "/root/reference/src/model.py"
OUT = os.path.join(ROOT, "tests", "golden")
N_CLIPS = 32
MARGIN_STD = 2.5        # class-margin spread of the golden head (what a trained detector produces)


def load_reference_model_module():
    spec = importlib.util.spec_from_file_location("ref_model", REF_MODEL)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    torch.manual_seed(20260227)

    # ---- features from the restated featuriser -----------------------------------
    wav = synth.make_clips(0, N_CLIPS)
    feats = featurizer.extract_features_batch(torch.from_numpy(wav))          # (12, 90, 101)
    crc = np.array([zlib.crc32(w.tobytes()) for w in wav], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "features_golden.npz"),
                        seeds=np.arange(N_CLIPS), wav_crc32=crc, features=feats.numpy())

    # ---- classifier goldens from the reference module itself ----------------------
    ref = load_reference_model_module()
    net = ref.create_model("residual", n_mels=90, num_classes=2, in_channels=1).eval()
    g = torch.Generator().manual_seed(7)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            c = m.num_features
            m.running_mean.copy_(torch.randn(c, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(c, generator=g) * 1.5 + 0.25)
            m.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(c, generator=g) * 0.2)
            m.num_batches_tracked.fill_(123)
    x = feats.unsqueeze(1).contiguous()                                       # (12, 1, 90, 101)
    with torch.no_grad():
        # head gain: scale fc.2 so the class margin has std MARGIN_STD over this input set, re-centre both logits
        # (a trained head has no 30-logit common-mode offset), then centre the margin so argmax sees both classes
        l = net(x)
        net.fc[2].weight.data.mul_(MARGIN_STD / (l[:, 1] - l[:, 0]).std())
        net.fc[2].bias.data.sub_(net(x).mean(dim=0))
        l = net(x)
        d = (l[:, 1] - l[:, 0]).sort().values
        net.fc[2].bias.data[1] -= 0.5 * (d[N_CLIPS // 2 - 1] + d[N_CLIPS // 2])
        a1 = net.conv1(x)
        a2 = net.res_blocks[0](a1)
        a3 = net.res_blocks[1](a2)
        logits = net(x)
        preds, probs = net.predict(x)
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "resnet_golden.npz"),
                        x=x.numpy(), a1=a1.numpy(), a2=a2.numpy(), a3=a3.numpy(),
                        logits=logits.numpy(), probs=probs.numpy(), preds=preds.numpy(),
                        **{"sd." + k: v for k, v in sd.items()})
    m = logits[:, 1] - logits[:, 0]
    print("margin std %.3f, min |margin| %.4f, max |logit| %.3f, max |fc.2.weight| %.2f"
          % (m.std(), m.abs().min(), logits.abs().max(), net.fc[2].weight.abs().max()))
    print("logits", logits.numpy().round(4).tolist())
    print("preds", preds.tolist())
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
