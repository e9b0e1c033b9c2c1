"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host mirror
keeps the reference's interface / error behaviour, and the streaming bookkeeping is right.
No GPU compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import cough_detector_amd as cda
from cough_detector_amd import _lib, _tables, preprocessing, synth
from oracle import featurizer as ofeat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIPPED = dict(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cough_amd.h")).read()
    declared = set(re.findall(r"\b(cough_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cough_amd_abi_version() == 5
    assert lib.cough_amd_arch() == b"gfx950"


def test_dynamic_symbol_table_is_exactly_the_header():
    """-fvisibility=hidden + a linker version script -- `nm -D` of the built library lists the entry points of
    include/cough_amd.h and nothing else (no C++-mangled internals, no std:: instantiations)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    header = open(os.path.join(ROOT, "include", "cough_amd.h")).read()
    declared = set(re.findall(r"\b(cough_[a-z_0-9]+)\s*\(", header))
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_ctypes_structs_match_header_layout():
    assert ctypes.sizeof(_lib.FeatConfig) == (14 + _lib.MAX_CONTRAST_BANDS + 2) * 4   # 14 scalars + contrast_edges[18]
    assert ctypes.sizeof(_lib.ConvBN) == 6 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.ResNetWeights) == (7 * 6 + 2) * 8 + 8


def test_argument_errors_need_no_gpu():
    lib = _lib.load()
    assert lib.cough_featurize(None, None, 16000, None, 1, 0, None) == _lib.EINVAL
    assert b"NULL" in lib.cough_amd_last_error()
    with pytest.raises(ValueError):
        _lib.check(lib.cough_resnet_create(None, None, 0), "x")
    # the ABI v5 entry points refuse NULL / negative arguments before anything touches a GPU
    assert lib.cough_featurize_any(None, None, 16000, 12000, None, 1, 0, None, 0, None) == _lib.EINVAL
    assert lib.cough_spectrogram_any(None, None, 16000, 12000, None, 1, 0, None) == _lib.EINVAL
    assert lib.cough_featurizer_num_frames_for(None, 16000) == -1 and lib.cough_featurizer_path(None) == -1
    assert lib.cough_featurizer_workspace_bytes_for(None, 16000, 4) == 0


def test_product_tables_equal_oracle_tables():
    assert torch.equal(_tables.hann_window(400), ofeat.hann_window())
    assert torch.equal(_tables.mel_filterbank(257, 100.0, 4000.0, 64, 16000), ofeat.melscale_fbanks())
    assert torch.equal(_tables.dct_matrix(13, 64), ofeat.create_dct())


def test_preprocessor_interface_and_errors():
    p = cda.AudioPreprocessor(**SHIPPED)
    assert p.get_num_features() == 90 and p.get_expected_time_frames() == 101
    assert cda.AudioPreprocessor(**{**SHIPPED, "use_delta_delta": True}).get_num_features() == 103
    with pytest.warns(UserWarning, match="NaN by construction"):
        full = cda.AudioPreprocessor()                       # reference defaults: PCEN, delta-delta, 6 contrast bands
    assert full.get_num_features() == 110                    # 64 + 3 * 13 + 7, README "110 features"
    assert cda.AudioPreprocessor(**{**SHIPPED, "use_mfcc": False}).get_num_features() == 64
    assert cda.AudioPreprocessor(**{**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 4}).get_num_features() == 95
    with pytest.raises(ValueError, match="n_contrast_bands"):
        cda.AudioPreprocessor(**{**SHIPPED, "use_spectral_contrast": True, "n_contrast_bands": 40})
    assert cda.AudioPreprocessor(use_spectral_contrast=False, use_delta_delta=False).get_num_features() == 90   # PCEN on
    with pytest.raises(ValueError, match="n_fft"):
        cda.AudioPreprocessor(n_fft=4096, **SHIPPED)
    with pytest.raises(ValueError, match="expected"):
        p.extract_features(torch.zeros(8000))                # (N,): the reference takes (1, N); any N is accepted
    same = torch.zeros(1, 16000)
    assert p.resample(same, 16000) is same                   # same rate: untouched, as the reference (:179-180)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            p.resample(torch.zeros(1, 44100), 44100)
    x = torch.arange(10.0).reshape(1, 10)
    assert torch.equal(p.pad_or_trim(x, 4), ofeat.pad_or_trim(x, 4))
    assert torch.equal(p.pad_or_trim(x, 15), ofeat.pad_or_trim(x, 15))
    z = torch.zeros(1, 5)
    if torch.cuda.is_available():
        assert torch.equal(p.normalize(z), z)               # all-zero input: unchanged, no 0 / 0 (:209-212)
    else:
        with pytest.raises(RuntimeError, match="no CPU fallback"):   # normalize / to_mono are kernels too
            p.normalize(z)
    assert p.to_mono(z) is z                                 # one channel: returned as is (:194-195)
    assert isinstance(cda.create_preprocessor(realtime=True, **SHIPPED), cda.RealtimePreprocessor)


def test_missing_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = cda.AudioPreprocessor(**SHIPPED)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p.extract_features(torch.zeros(1, 16000))
    m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 90, 101))


def test_realtime_window_bookkeeping_matches_oracle():
    rt = cda.RealtimePreprocessor(window_duration=1.0, hop_duration=0.25, **SHIPPED)
    ow = ofeat.RealtimeWindowerOracle(1.0, 0.25)
    stream = torch.from_numpy(synth.make_stream(3, 3.3))
    rng = np.random.default_rng(0)
    pos, got = 0, []
    oracle_windows = []
    while pos < stream.numel():
        n = int(rng.choice([1600, 1600, 100, 7000, 20000]))
        chunk = stream[pos:pos + n]
        pos += n
        w = rt.take_windows(chunk)
        if w is not None:
            got.extend(w)
        # the oracle's own loop, capturing the raw windows it would featurise
        ow.buffer = torch.cat([ow.buffer, chunk[None]], dim=1)
        while ow.buffer.shape[1] >= 16000:
            oracle_windows.append(ow.buffer[0, :16000].clone())
            ow.buffer = ow.buffer[:, 4000:]
        assert rt.buffer.shape == ow.buffer.shape
    assert len(got) == len(oracle_windows) > 5
    for a, b in zip(got, oracle_windows):
        assert torch.equal(a, b)
    rt.reset()
    assert rt.buffer.shape == (1, 0)


def test_model_state_dict_contract(resnet_golden):
    sd, _ = resnet_golden
    m = cda.create_model("residual", n_mels=90, num_classes=2, in_channels=1)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd)                                    # strict
    assert cda.count_parameters(m) == 290370
    with pytest.raises(ValueError, match="Unknown model type"):
        cda.create_model("resnet50")
    # the other two reference models: same factory, reference state_dict keys (goldens from the reference modules)
    assert isinstance(cda.create_model("small"), cda.CoughDetectorSmall)
    assert isinstance(cda.create_model(), cda.CoughDetector)                 # the factory default is "standard"
    m.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        m(torch.zeros(1, 1, 90, 101))


def test_conv_stack_models_take_the_reference_state_dicts(cnn_golden):
    for kind, n_params in (("standard", 421954), ("small", 21122)):
        sd, _ = cnn_golden[kind]
        m = cda.create_model(kind, n_mels=90, num_classes=2, in_channels=1, compute_dtype="bf16_approx")
        assert set(m.state_dict().keys()) == set(sd.keys())
        m.load_state_dict(sd)                                                # strict
        assert cda.count_parameters(m) == n_params
    with pytest.raises(ValueError, match="in_channels"):
        cda.CoughDetector(in_channels=3)
    with pytest.raises(ValueError, match="compute_dtype"):
        cda.CoughDetectorSmall(compute_dtype="fp8")


def test_num_features_from_config():
    from cough_detector_amd.inference import num_features_from_config
    shipped = dict(n_mels=64, n_mfcc=13, use_mfcc=True, use_delta_delta=False, use_spectral_contrast=False)
    assert num_features_from_config(shipped) == 90
    assert num_features_from_config({}) == 110             # the reference's fall-back defaults


def test_spec_augment_draw_order_matches_oracle():
    """Host half of SpecAugment (no GPU): the coin and the mask draws consume the RNGs exactly as the restatement."""
    import random
    from cough_detector_amd.augmentation import SpecAugment
    from oracle import augmentation as oaug
    aug = SpecAugment(freq_mask_param=10, time_mask_param=20, n_freq_masks=2, n_time_masks=2, p=1.0)
    torch.manual_seed(11)
    masks = aug.draw_masks(90, 101)
    torch.manual_seed(11)
    want = [(0,) + oaug.draw_mask(10, 90) for _ in range(2)] + [(1,) + oaug.draw_mask(20, 101) for _ in range(2)]
    assert masks == want and all(e - s < (10 if a == 0 else 20) for a, s, e in masks)
    x = torch.ones(1, 4, 4)
    random.seed(0)
    assert SpecAugment(p=0.0)(x) is x                       # the coin (python `random`) says no: input returned as is


def test_visible_gpu_count_reads_sysfs_without_the_runtime(tmp_path):
    """bench.py's multi-rank parent counts devices from the KFD topology (no HIP call): CPU nodes are skipped,
    *_VISIBLE_DEVICES and the render nodes this process may open cut the count down."""
    from cough_detector_amd.hostcpu import visible_gpu_count
    nodes = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):                     # 2 CPU sockets + 3 GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    dri = tmp_path / "dri"
    dri.mkdir()
    assert visible_gpu_count(str(nodes), environ={}, dri_root=str(tmp_path / "missing")) == 3
    assert visible_gpu_count(str(nodes), environ={"HIP_VISIBLE_DEVICES": "0,2"}, dri_root=str(tmp_path / "missing")) == 2
    assert visible_gpu_count(str(nodes), environ={"ROCR_VISIBLE_DEVICES": ""}, dri_root=str(tmp_path / "missing")) == 0
    (dri / "renderD128").write_text("")
    (dri / "card0").write_text("")
    assert visible_gpu_count(str(nodes), environ={}, dri_root=str(dri)) == 1            # one render node handed in
    assert visible_gpu_count(str(tmp_path / "absent"), environ={}) is None


def _fake_node(tmp_path, gpus):
    """A sysfs look-alike: 2 CPU nodes + GPUs [(render minor, pci, numa node)], plus NUMA cpulists."""
    nodes, drm, numa = tmp_path / "nodes", tmp_path / "drm", tmp_path / "node"
    for i in range(2):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    for j, (minor, pci, node) in enumerate(gpus):
        d = nodes / str(2 + j)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\n")
        pdev = tmp_path / "pci" / pci
        pdev.mkdir(parents=True)
        (pdev / "numa_node").write_text(f"{node}\n")
        (drm / f"renderD{minor}").mkdir(parents=True)
        os.symlink(pdev, drm / f"renderD{minor}" / "device")
    return dict(kfd_root=str(nodes), drm_root=str(drm), dri_root=str(tmp_path / "no_dri")), numa


def test_gpu_numa_node_follows_kfd_order_and_visible_devices(tmp_path):
    """rank -> GPU -> PCI function -> NUMA node from sysfs only (no runtime call)."""
    from cough_detector_amd.hostcpu import explicit_device_limit, gpu_numa_node
    sysfs, _ = _fake_node(tmp_path, [(128, "0000:05:00.0", 0), (129, "0000:15:00.0", 0), (130, "0000:85:00.0", 1),
                                     (131, "0000:95:00.0", -1)])
    assert gpu_numa_node(0, environ={}, **sysfs) == (0, "0000:05:00.0")
    assert gpu_numa_node(2, environ={}, **sysfs) == (1, "0000:85:00.0")
    assert gpu_numa_node(3, environ={}, **sysfs) == (None, "0000:95:00.0")        # numa_node -1: unknown
    assert gpu_numa_node(4, environ={}, **sysfs) == (None, None)                  # no such device
    assert gpu_numa_node(0, environ={"HIP_VISIBLE_DEVICES": "2,0"}, **sysfs) == (1, "0000:85:00.0")
    assert gpu_numa_node(0, environ={"ROCR_VISIBLE_DEVICES": "1,2", "HIP_VISIBLE_DEVICES": "1"}, **sysfs) == (1, "0000:85:00.0")
    assert gpu_numa_node(0, environ={"ROCR_VISIBLE_DEVICES": "GPU-deadbeef"}, **sysfs) == (None, None)
    assert explicit_device_limit({}) is None and explicit_device_limit({"HIP_VISIBLE_DEVICES": "0,1,2"}) == 3
    assert explicit_device_limit({"ROCR_VISIBLE_DEVICES": "0,1", "HIP_VISIBLE_DEVICES": "0,1,2"}) == 2
    assert explicit_device_limit({"CUDA_VISIBLE_DEVICES": ""}) == 0


def test_bind_to_gpu_numa_sets_the_affinity_of_the_node(tmp_path):
    from cough_detector_amd.hostcpu import bind_to_gpu_numa
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no affinity call on this platform")
    before = os.sched_getaffinity(0)
    cpus = sorted(before)
    sysfs, numa = _fake_node(tmp_path, [(128, "0000:05:00.0", 0), (129, "0000:85:00.0", 1)])
    (numa / "node0").mkdir(parents=True)
    (numa / "node1").mkdir(parents=True)
    half = cpus[:max(1, len(cpus) // 2)]
    (numa / "node0" / "cpulist").write_text(f"{half[0]}-{half[-1]}\n")
    (numa / "node1" / "cpulist").write_text("100000-100007\n")                   # CPUs this process may not use
    try:
        info = bind_to_gpu_numa(0, node_root=str(numa), environ={}, **sysfs)
        assert info == {"numa_node": 0, "pci": "0000:05:00.0", "cpus": len(set(range(half[0], half[-1] + 1)) & before)}
        assert os.sched_getaffinity(0) == set(range(half[0], half[-1] + 1)) & before
        os.sched_setaffinity(0, before)
        info = bind_to_gpu_numa(1, node_root=str(numa), environ={}, **sysfs)      # empty intersection: left alone
        assert info["numa_node"] is None and os.sched_getaffinity(0) == before
        info = bind_to_gpu_numa(5, node_root=str(numa), environ={}, **sysfs)      # unknown device: left alone
        assert info == {"numa_node": None, "pci": None, "cpus": None} and os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)


def test_rebind_to_the_numa_node_of_the_device_the_runtime_really_gave(tmp_path):
    """bench.py checks the PCI address of its device after init: a rank that was bound by sysfs order to the wrong GPU's
    NUMA node moves to the right one (and the first binding must not have narrowed what it may move to)."""
    from cough_detector_amd import hostcpu
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no affinity call on this platform")
    before = os.sched_getaffinity(0)
    cpus = sorted(before)
    if len(cpus) < 2:
        pytest.skip("needs two CPUs")
    lo, hi = cpus[:len(cpus) // 2], cpus[len(cpus) // 2:]
    sysfs, numa = _fake_node(tmp_path, [(128, "0000:05:00.0", 0), (129, "0000:85:00.0", 1)])
    for k, half in ((0, lo), (1, hi)):
        (numa / f"node{k}").mkdir(parents=True)
        (numa / f"node{k}" / "cpulist").write_text(",".join(str(c) for c in half) + "\n")
    try:
        info = hostcpu.bind_to_gpu_numa(0, node_root=str(numa), environ={}, **sysfs)
        assert info["numa_node"] == 0 and os.sched_getaffinity(0) == set(lo)
        same = hostcpu.rebind_to_pci_numa("0000:05:00.0", info, node_root=str(numa), pci_root=str(tmp_path / "pci"))
        assert same["rebound"] is False and os.sched_getaffinity(0) == set(lo)
        moved = hostcpu.rebind_to_pci_numa("0000:85:00.0", info, node_root=str(numa), pci_root=str(tmp_path / "pci"))
        assert moved["rebound"] is True and moved["numa_node"] == 1 and moved["pci"] == "0000:85:00.0"
        assert os.sched_getaffinity(0) == set(hi)
    finally:
        os.sched_setaffinity(0, before)


def test_effective_dtype_reports_the_kernels_that_run():
    """A reduced-precision request the compiled kernels do not cover is reported (and warned about at the first
    forward, tests/test_gpu_fuzz.py), not silently replaced."""
    m = cda.create_model("residual", n_mels=90, compute_dtype="bf16x3")
    assert m.compute_dtype == "bf16x3" and m.effective_dtype() == "bf16x3" and m.effective_dtype(80, 101) == "fp32"
    assert m.effective_dtype(103, 101) == "bf16x3" and m.effective_dtype(110, 101) == "bf16x3"     # the reference's own flags
    assert m.effective_dtype(100, 101) == "fp32" and m.effective_dtype(103, 201) == "fp32"
    assert all(m.effective_dtype(h, 101) == "bf16x3" for h in (64, 66, 67, 68, 92, 93, 94, 95))   # use_mfcc=False / 1..4 contrast bands
    # the edges of the compiled ranges (stem output rows ((h - 1) // 2 + 1) // 2 in 16, 17 | 22..24 | 26, 27): 63..70, 87..98, 103..110
    edges = {62: "fp32", 63: "bf16x3", 66: "bf16x3", 67: "bf16x3", 70: "bf16x3", 71: "fp32", 86: "fp32", 87: "bf16x3", 98: "bf16x3",
             99: "fp32", 102: "fp32", 103: "bf16x3", 110: "bf16x3", 111: "fp32"}
    assert {h: m.effective_dtype(h, 101) for h in edges} == edges
    assert all(m.effective_dtype(h, 101) == "bf16x3" for h in (105, 106, 107, 108, 109))   # + 1..5 contrast bands
    wide = cda.CoughDetectorResidual(channels=(16, 24, 40), compute_dtype="bf16x3")
    assert wide.effective_dtype() == "fp32" and wide.compute_dtype == "bf16x3"
    assert cda.create_model("residual", compute_dtype="fp32").effective_dtype(33, 77) == "fp32"
    assert cda.create_model("residual", compute_dtype="bf16_approx").effective_dtype() == "bf16_approx"
    with pytest.warns(UserWarning, match="APPROXIMATE single-bf16 mode"):      # the old name still works, loudly
        assert cda.create_model("residual", compute_dtype="bf16").compute_dtype == "bf16_approx"


def test_featuriser_refuses_non_tensor_and_integer_waveforms():
    import numpy as np
    pre = cda.AudioPreprocessor(use_pcen=False, use_pre_emphasis=False, use_delta_delta=False, use_spectral_contrast=False)
    with pytest.raises(TypeError, match="torch.Tensor"):
        pre.extract_features(np.zeros((1, 16000), np.float32))
    with pytest.raises(TypeError, match="floating-point"):
        pre.extract_features(torch.zeros(1, 16000, dtype=torch.int16))     # torch.stft refuses integer input as well


def test_probability_logs_are_bounded_and_a_zero_hop_is_refused():
    """a detector that runs for months must not grow (one float per window forever), and
    hop_duration <= 0 is an endless loop in the reference's add_audio -- a ValueError here."""
    log = preprocessing.RecentLog(5)
    for i in range(1000):
        log.append(i)
        assert len(log) < 10
    assert list(log)[-5:] == [995, 996, 997, 998, 999] and log[-1] == 999 and log[-3:] == [997, 998, 999]
    with pytest.raises(ValueError):
        preprocessing.RecentLog(0)
    for hop in (0.0, -0.25, 1e-6):
        with pytest.raises(ValueError, match="hop_duration"):
            cda.RealtimePreprocessor(window_duration=1.0, hop_duration=hop, **SHIPPED)
    rt = cda.RealtimePreprocessor(window_duration=1.0, hop_duration=1.0 / 16000, **SHIPPED)
    assert rt.hop_samples == 1


def test_load_audio_reads_wave_files_like_torchaudio_load(tmp_path):
    """AudioPreprocessor.load_audio = torchaudio.load(path) (/root/reference/src/preprocessing.py:155-166) for WAVE files:
    (channels, samples) float32 scaled to [-1, 1) and the file's sample rate; anything else is refused with a ValueError."""
    from scipy.io import wavfile
    pre = cda.AudioPreprocessor(**SHIPPED)
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.9, 0.9, size=(4410, 2)).astype(np.float32)
    cases = {"f32": (x, x), "i16": ((x * 32767).astype(np.int16), None), "u8": ((x * 127 + 128).astype(np.uint8), None),
             "i32": ((x.astype(np.float64) * 2147483647).astype(np.int32), None)}
    for name, (data, want) in cases.items():
        path = str(tmp_path / f"{name}.wav")
        wavfile.write(path, 44100, data)
        w, sr = pre.load_audio(path)
        assert sr == 44100 and w.shape == (2, 4410) and w.dtype == torch.float32 and w.is_contiguous()
        if want is None:
            scale = {"i16": 32768.0, "u8": 128.0, "i32": 2147483648.0}[name]
            want = (data.astype(np.float64) - (128 if name == "u8" else 0)) / scale
        assert np.abs(w.numpy().T - want).max() < 1e-6 and np.abs(w.numpy().T - x).max() < 2e-2      # (8-bit: 1 / 128 steps)
    mono = str(tmp_path / "mono.wav")
    wavfile.write(mono, 16000, (x[:, 0] * 32767).astype(np.int16))
    w, sr = pre.load_audio(mono)
    assert w.shape == (1, 4410) and sr == 16000
    bad = tmp_path / "not_audio.wav"
    bad.write_bytes(b"ID3 definitely not RIFF")
    with pytest.raises(ValueError, match="load_audio"):
        pre.load_audio(str(bad))
