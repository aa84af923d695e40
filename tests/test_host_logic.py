"""CPU: host-side logic — config parsing, state-dict layout, C-ABI exports, sharding over gloo."""
import ctypes
import os
import re
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests.util import manifest, ROOT


def test_yaml_configs_parse(tmp_path):
    from wavtokenizer_amd.config import arch_from_yaml_dict, ARCH_HOP600, ARCH_HOP320
    for arch in (ARCH_HOP600, ARCH_HOP320):
        node = {"model": {"init_args": {
            "feature_extractor": {"class_path": "decoder.feature_extractors.EncodecFeatures",
                                  "init_args": {"encodec_model": "encodec_24khz", "bandwidths": list(arch.bandwidths),
                                                "num_quantizers": 1, "dowmsamples": list(arch.ratios), "vq_bins": 4096}},
            "backbone": {"class_path": "decoder.models.VocosBackbone",
                         "init_args": {"input_channels": 512, "dim": 768, "intermediate_dim": 2304, "num_layers": 12,
                                       "adanorm_num_embeddings": 4}},
            "head": {"class_path": "decoder.heads.ISTFTHead",
                     "init_args": {"dim": 768, "n_fft": arch.n_fft, "hop_length": arch.hop_length, "padding": "same"}}}}}
        assert arch_from_yaml_dict(node) == arch
        assert arch.hop == arch.hop_length
    assert ARCH_HOP600.frames(72000) == 120 and ARCH_HOP320.frames(61920) == 194      # wavtokenizer.txt:157
    bad = {"model": {"init_args": {"feature_extractor": {"class_path": "x.MelSpectrogramFeatures"}, "backbone": {}, "head": {}}}}
    with pytest.raises(ValueError):
        arch_from_yaml_dict(bad)


@pytest.mark.parametrize("name", ["hop600", "hop320"])
def test_state_dict_layout_matches_reference(name):
    """Keys and shapes equal the reference module's own state_dict (dumped into the manifest)."""
    from wavtokenizer_amd.state_spec import full_state_spec
    from wavtokenizer_amd.config import NAMED_ARCHS
    want = manifest()["archs"][name]["ref_state_keys"]
    got = {k: list(v) for k, v in full_state_spec(NAMED_ARCHS[name]).items()}
    assert got == want


def test_library_exports_the_whole_c_abi():
    """libwavtok_hip.so loads and exports every function include/wavtokenizer_amd.h declares."""
    from wavtokenizer_amd import _capi
    header = open(os.path.join(ROOT, "include", "wavtokenizer_amd.h")).read()
    declared = set(re.findall(r"\b(wt_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    for sym in declared:
        assert getattr(_capi.lib, sym) is not None
    assert b"gfx950" in _capi.lib.wt_version()


def test_module_tree_and_cpu_refusal():
    from wavtokenizer_amd import WavTokenizer, ARCH_HOP600
    m = WavTokenizer.from_arch(ARCH_HOP600)
    fe = m.feature_extractor
    assert fe.encodec.quantizer.bins == 4096 and fe.bandwidths == [6.6, 6.6, 6.6, 6.6]
    assert tuple(fe.encodec.quantizer.vq.layers[0].codebook.shape) == (4096, 512)
    assert callable(fe.encodec.encoder) and callable(fe.encodec.decoder) and callable(m.backbone)
    assert len(m.state_dict()) == 289
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.encode_infer(torch.zeros(1, 600), bandwidth_id=torch.tensor([0]))


def test_shard_bounds_cover_everything():
    from wavtokenizer_amd.sharding import shard_bounds
    for n in (1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeCodec:
    """Stands in for the GPU model in the gloo test: per-clip, deterministic, batch-independent."""
    def encode_infer(self, wav, bandwidth_id=None):
        codes = (wav[:, ::600].abs() * 4095).long().clamp_(0, 4095).unsqueeze(0)
        return wav[:, None, ::600].repeat(1, 512, 1), codes

    def decode(self, feats, bandwidth_id=None):
        return feats[:, 0, :].repeat_interleave(600, dim=1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, q):
    import torch.distributed as dist
    from wavtokenizer_amd.sharding import roundtrip_sharded
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    wav = torch.rand(n_clips, 3000, generator=g)
    codes, out = roundtrip_sharded(_FakeCodec(), wav, torch.tensor([0]), dist, rank, world, dst=0)
    m = _FakeCodec()
    f, c = m.encode_infer(wav)
    ok = torch.equal(codes, c) and (rank != 0 or torch.equal(out, m.decode(f))) and (rank == 0 or out is None)
    # the non-blocking form used by bench.py gives the same result
    from wavtokenizer_amd.sharding import gather_async, shard_bounds
    lo, hi = shard_bounds(n_clips, rank, world)
    counts = [shard_bounds(n_clips, r, world)[1] - shard_bounds(n_clips, r, world)[0] for r in range(world)]
    fl, cl = m.encode_infer(wav[lo:hi])
    codes2, out2 = gather_async(cl, m.decode(fl), dist, world, rank, dst=0, counts=counts).result()
    ok = ok and torch.equal(codes2, c) and (rank != 0 or torch.equal(out2, m.decode(f))) and (rank == 0 or out2 is None)
    q.put((rank, bool(ok), tuple(codes.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [8, 7])
def test_sharded_roundtrip_world2_gloo(n_clips):
    """N > 1 path: two ranks over gloo, even and uneven shards; the gathers rebuild the clip order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (1, n_clips, 5) for _, _, shape in res)


def _fake_lightning_ckpt(path, arch_name, scale=1.0):
    """A Lightning-style checkpoint: {'state_dict': {...}} with discriminator keys the loader must drop
    (decoder/pretrained.py:101-105)."""
    from wavtokenizer_amd.state_spec import full_state_spec
    from wavtokenizer_amd.config import NAMED_ARCHS
    g = torch.Generator().manual_seed(3)
    sd = {k: torch.randn(shape, generator=g) * 0.01 * scale for k, shape in full_state_spec(NAMED_ARCHS[arch_name]).items()}
    sd["feature_extractor.encodec.quantizer.vq.layers.0._codebook.inited"] = torch.ones(1)
    sd["multiperioddisc.discriminators.0.convs.0.weight"] = torch.zeros(4, 1, 5, 1)
    sd["melspec_loss.mel_spec.spectrogram.window"] = torch.zeros(16)
    torch.save({"state_dict": sd, "epoch": 3, "global_step": 10}, path)
    return sd


def _yaml_for(arch_name, tmp_path):
    import yaml
    from wavtokenizer_amd.config import NAMED_ARCHS
    arch = NAMED_ARCHS[arch_name]
    cfg = {"model": {"class_path": "decoder.experiment.WavTokenizer", "init_args": {
        "sample_rate": 24000,
        "feature_extractor": {"class_path": "decoder.feature_extractors.EncodecFeatures",
                              "init_args": {"encodec_model": "encodec_24khz", "bandwidths": [6.6, 6.6, 6.6, 6.6],
                                            "train_codebooks": True, "num_quantizers": 1,
                                            "dowmsamples": list(arch.ratios), "vq_bins": 4096, "vq_kmeans": 200}},
        "backbone": {"class_path": "decoder.models.VocosBackbone",
                     "init_args": {"input_channels": 512, "dim": 768, "intermediate_dim": 2304, "num_layers": 12,
                                   "adanorm_num_embeddings": 4}},
        "head": {"class_path": "decoder.heads.ISTFTHead",
                 "init_args": {"dim": 768, "n_fft": arch.n_fft, "hop_length": arch.hop_length, "padding": "same"}}}}}
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    return str(p)


def test_from_pretrained0802_filters_and_loads(tmp_path):
    from wavtokenizer_amd import WavTokenizer
    cfg = _yaml_for("hop320", tmp_path)
    ck = tmp_path / "m.ckpt"
    sd = _fake_lightning_ckpt(str(ck), "hop320")
    m = WavTokenizer.from_pretrained0802(cfg, str(ck))
    assert not m.training and m.arch.hop == 320
    got = m.state_dict()
    assert len(got) == 289 and not any(k.startswith("multiperioddisc") for k in got)
    for k in ("backbone.convnext.3.pwconv1.weight", "feature_extractor.encodec.encoder.model.3.conv.conv.weight_v", "head.out.bias"):
        assert torch.equal(got[k], sd[k])
    # a checkpoint with a missing key is refused like the reference's strict load_state_dict
    sd2 = {k: v for k, v in sd.items() if k != "head.out.bias"}
    torch.save({"state_dict": sd2}, str(ck))
    with pytest.raises(RuntimeError, match="Missing key"):
        WavTokenizer.from_pretrained0802(cfg, str(ck))


def test_from_pretrained0911_averages_best_three(tmp_path):
    """pretrained.py:117-156: mean of the three `vocos_*` checkpoints with the smallest val-loss suffix."""
    from wavtokenizer_amd import WavTokenizer
    cfg = _yaml_for("hop600", tmp_path)
    folder = tmp_path / "ckpts"
    folder.mkdir()
    losses = {"vocos_checkpoint_epoch=1_step=1_val_loss=5.1000.ckpt": 1.0, "vocos_checkpoint_epoch=2_step=2_val_loss=4.9000.ckpt": 2.0,
              "vocos_checkpoint_epoch=3_step=3_val_loss=5.0000.ckpt": 3.0, "vocos_checkpoint_epoch=4_step=4_val_loss=6.0000.ckpt": 100.0,
              "last.ckpt": 50.0}
    sds = {n: _fake_lightning_ckpt(str(folder / n), "hop600", scale=s) for n, s in losses.items()}
    m = WavTokenizer.from_pretrained0911(cfg, str(folder))
    key = "backbone.embed.bias"
    best = sorted(n for n in losses if n.startswith("vocos_"))          # names sort by their val-loss suffix here
    picked = sorted((n for n in losses if n.startswith("vocos_")), key=lambda n: n[-11:-5])[:3]
    want = sum(sds[n][key] for n in picked) / 3
    assert torch.allclose(m.state_dict()[key], want, rtol=1e-6, atol=1e-9), (best, picked)


def test_packed_weights_round_trip(tmp_path):
    """save_packed / from_packed: one safetensors file with exactly the hot-path keys, nothing
    executed on load, values bit-identical."""
    from safetensors import safe_open
    from wavtokenizer_amd import WavTokenizer
    cfg = _yaml_for("hop600", tmp_path)
    sd = _fake_lightning_ckpt(str(tmp_path / "m.ckpt"), "hop600")
    m = WavTokenizer.from_pretrained0802(cfg, str(tmp_path / "m.ckpt"))
    path = str(tmp_path / "hot_path.safetensors")
    m.save_hot_state(path)
    with safe_open(path, framework="pt") as f:
        assert set(f.keys()) == set(m.state_dict().keys()) and len(list(f.keys())) == 289
        assert f.metadata()["format"] == "wavtokenizer_amd.hot_state.v1"
    m2 = WavTokenizer.from_hot_state(cfg, path)
    assert not m2.training
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v), k


def test_graph_plan_selection_is_host_logic():
    """Which calls get a hipGraph plan (WT_PLAN_FLAG_GRAPH) is decided on the host: batches up to the limit, never with
    debug taps (stage buffers must stay addressable), off when the limit is 0; the flag is part of the plan key."""
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS, _capi
    m = WavTokenizer.from_arch(NAMED_ARCHS["hop600"])
    assert m._graph_flags(1) & _capi.WT_PLAN_FLAG_GRAPH and m._graph_flags(16) & _capi.WT_PLAN_FLAG_GRAPH
    assert not (m._graph_flags(17) & _capi.WT_PLAN_FLAG_GRAPH)
    m.set_graph_max_clips(64)
    assert m._graph_flags(64) & _capi.WT_PLAN_FLAG_GRAPH
    m.set_graph_max_clips(0)
    assert m._graph_flags(1) == 0
    m.set_graph_max_clips(16)
    m.set_debug_keep_stages(True)
    assert m._graph_flags(1) == _capi.WT_PLAN_FLAG_KEEP_STAGES
    m.set_debug_keep_stages(False)
    m.set_lstm_mode("step")
    assert m._graph_flags(2) == (_capi.WT_PLAN_FLAG_STEP_LSTM | _capi.WT_PLAN_FLAG_GRAPH)
    assert {_capi.WT_PLAN_FLAG_KEEP_STAGES, _capi.WT_PLAN_FLAG_FP32_GEMM, _capi.WT_PLAN_FLAG_STEP_LSTM,
            _capi.WT_PLAN_FLAG_GRAPH} == {1, 2, 4, 8}


def test_packed_image_header_is_validated_without_a_gpu():
    """wt_packed_info (the header check of from_packed): magic, layout version, architecture hash, truncation."""
    import ctypes
    import numpy as np
    from wavtokenizer_amd import _capi
    lib = _capi.lib

    def info(buf):
        wa, ver, ah = _capi.WtArch(), ctypes.c_int32(), ctypes.c_uint64()
        rc = lib.wt_packed_info(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, ctypes.byref(wa), ctypes.byref(ver), ctypes.byref(ah))
        return rc, lib.wt_last_error().decode()

    rc, msg = info(np.zeros(16, np.uint8))
    assert rc != 0 and "short" in msg
    junk = np.frombuffer(b"NOPE" + bytes(4096), dtype=np.uint8).copy()
    rc, msg = info(junk)
    assert rc != 0 and "magic" in msg
    # a header with the right magic but another layout version
    hdr = np.zeros(4096, np.uint8)
    hdr[:4] = np.frombuffer(b"WTPK", dtype=np.uint8)
    hdr[4:8] = np.frombuffer(np.int32(1).tobytes(), dtype=np.uint8)
    rc, msg = info(hdr)
    assert rc != 0 and "version" in msg
    import re
    current = int(re.search(r"this library reads (\d+)", msg).group(1))
    hdr[4:8] = np.frombuffer(np.int32(current).tobytes(), dtype=np.uint8)
    rc, msg = info(hdr)
    assert rc != 0 and "hash" in msg                   # zeroed architecture + zero hash: caught as a corrupt header


def test_bench_finds_the_dominant_kernel_in_the_committed_pmc_summary():
    """roofline.traffic comes from profiles/r*_pmc_traffic.json; a renamed kernel (new template parameter) must not turn it
    into null unnoticed."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = bench.pmc_traffic("hop600", 64, 3)
    assert t is not None and 101e6 < t < 400e6, t


# ------------------------------------------------------------------------------------------ round 3
class _RecordingCodec:
    """Deterministic stand-in for the GPU model in the world-2 test of the step runner: per-clip, batch-independent
    results, and every enqueue is visible in the runner's log (StepRunner logs encode / issue / decode / collect)."""
    hop = 600

    def encode_infer(self, wav, bandwidth_id=None):
        codes = (wav[:, ::self.hop].abs() * 4095).long().clamp_(0, 4095).unsqueeze(0)
        return wav[:, None, ::self.hop].repeat(1, 512, 1), codes

    def decode(self, feats, bandwidth_id=None):
        return feats[:, 0, :].repeat_interleave(self.hop, dim=1)


def _runner_worker(rank, world, port, n_steps, q):
    import torch.distributed as dist
    from wavtokenizer_amd.sharding import StepRunner, check_exchange_order, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    wav_all = torch.rand(6, 3000, generator=g)
    lo, hi = shard_bounds(6, rank, world)
    log = []
    model = _RecordingCodec()
    r = StepRunner(model, wav_all[lo:hi].contiguous(), torch.tensor([0]), dist, world, rank, gather=True, backend="gloo", log=log)
    results = []
    for _ in range(n_steps):
        _c, _o, res = r.step()
        if res is not None:
            results.append(res)
    results.append(r.drain())
    ok = True
    try:
        check_exchange_order(log, n_steps)
    except AssertionError as e:
        ok = False
        log.append(("order-violation", str(e)))
    f, c = model.encode_infer(wav_all)
    want_wav = model.decode(f)
    for codes, wav in results:                 # every exchange delivers all clips in clip order
        ok = ok and torch.equal(codes, c) and (rank != 0 or torch.equal(wav, want_wav)) and (rank == 0 or wav is None)
    ok = ok and len(results) == n_steps and r.exchanges == n_steps
    q.put((rank, bool(ok), log))
    dist.barrier()
    dist.destroy_process_group()


def test_step_runner_exchange_order_world2_gloo():
    """The ordering argument of bench.py --gpus N, asserted instead of rehearsed by hand: the exchange of
    step i is issued after step i+1's encode has been enqueued and collected before step i+2's encode, so RCCL's kernels
    can never run beside the persistent LSTM; every exchange delivers all clips in order; the last one is drained."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_steps = 4
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, port, n_steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    # the invariant checker itself rejects the wrong order (an exchange issued before the next encode)
    from wavtokenizer_amd.sharding import check_exchange_order
    bad = [("encode", 0), ("decode", 0), ("issue", 0), ("encode", 1), ("decode", 1), ("collect", 0), ("issue", 1), ("collect", 1)]
    with pytest.raises(AssertionError):
        check_exchange_order(bad, 2)


def _packed_header(arch, version=None, n_allocs=0, struct_bytes=0, payload_bytes=0, total=4096):
    """A packed-image header as weights.cpp lays it out (magic, version, wt_arch, FNV-1a hash of both, sizes, body hash)."""
    import numpy as np
    from wavtokenizer_amd import _capi
    wa = _capi.WtArch()
    wa.n_ratios = len(arch.ratios)
    for i, r in enumerate(arch.ratios):
        wa.ratios[i] = r
    wa.vq_bins, wa.num_quantizers, wa.input_channels = arch.vq_bins, arch.num_quantizers, arch.input_channels
    wa.dim, wa.intermediate_dim, wa.num_layers = arch.dim, arch.intermediate_dim, arch.num_layers
    wa.adanorm_num_embeddings, wa.n_fft, wa.hop_length = arch.adanorm_num_embeddings, arch.n_fft, arch.hop_length
    wa.padding_same = 1 if arch.padding == "same" else 0
    if version is None:          # ask the library which layout version it reads
        probe = np.zeros(4096, np.uint8)
        probe[:4] = np.frombuffer(b"WTPK", dtype=np.uint8)
        probe[4:8] = np.frombuffer(np.int32(1).tobytes(), dtype=np.uint8)
        _capi.lib.wt_packed_info(probe.ctypes.data_as(ctypes.c_void_p), probe.nbytes, None, None, None)
        version = int(re.search(r"this library reads (\d+)", _capi.lib.wt_last_error().decode()).group(1))
    arch_bytes = bytes(wa)
    h = 1469598103934665603
    for byte in arch_bytes + np.int32(version).tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    buf = np.zeros(total, np.uint8)
    hdr = b"WTPK" + np.int32(version).tobytes() + arch_bytes
    hdr += bytes((-len(hdr)) % 8)
    hdr += np.array([h, n_allocs, struct_bytes, payload_bytes, 0], dtype=np.uint64).tobytes()
    buf[:len(hdr)] = np.frombuffer(hdr, dtype=np.uint8)
    return buf


def test_from_packed_rejects_any_architecture_mismatch(tmp_path):
    """The C side sizes its launches from the image's wt_arch, the Python class its tensors from the YAML: a 'same' image
    under a 'center' config would be written past the end of the waveform tensor.  Every field is compared (no GPU needed:
    the check runs on the header before anything is uploaded)."""
    import dataclasses
    import yaml
    from wavtokenizer_amd import WavTokenizer, NAMED_ARCHS
    base = NAMED_ARCHS["hop600"]
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(yaml.safe_dump({"model": {"init_args": base.to_yaml_node()}}))
    for field, value in (("padding", "center"), ("adanorm_num_embeddings", 2), ("num_layers", 8), ("vq_bins", 1024)):
        other = dataclasses.replace(base, **{field: value})
        img = tmp_path / f"{field}.wtpk"
        _packed_header(other).tofile(str(img))
        with pytest.raises(ValueError, match=field):
            WavTokenizer.from_packed(str(cfg), str(img))


def test_packed_header_sizes_cannot_wrap():
    """Header sizes are attacker-controlled: huge values must read as 'truncated', never wrap around size_t."""
    from wavtokenizer_amd import _capi, NAMED_ARCHS
    lib = _capi.lib
    arch = NAMED_ARCHS["hop600"]

    def info(buf):
        rc = lib.wt_packed_info(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, None, None, None)
        return rc, lib.wt_last_error().decode()

    assert info(_packed_header(arch))[0] == 0                                   # a well-formed empty header passes the header check
    for kw in ({"struct_bytes": 2 ** 64 - 64}, {"struct_bytes": 2 ** 63}, {"n_allocs": 2 ** 61}, {"n_allocs": 2 ** 64 - 1},
               {"payload_bytes": 2 ** 64 - 256}, {"payload_bytes": 10 ** 9}, {"n_allocs": 3, "struct_bytes": 4096}):
        rc, msg = info(_packed_header(arch, **kw))
        assert rc != 0 and "truncated" in msg, (kw, msg)
    # the full check also hashes the body: a header-only image with a zero hash field is refused
    buf = _packed_header(arch)
    assert lib.wt_packed_verify(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) != 0
    assert "hash" in lib.wt_last_error().decode()


def test_no_kernel_spills_or_scratch():
    """Code-object metadata of the built library (tools/kernel_resources.py reads the notes of every gfx950 kernel in the
    .so; no GPU needed): no kernel may spill vector registers or use scratch memory.  Round 2 shipped a stage-1 kernel with
    16 spilled VGPRs (hipcc kept 80 raw weight registers alive across the tile loop for a maximum taken at the end) and
    reachable gemm16s / step-LSTM instantiations with 8-12: nothing in the build noticed.  One exception, by name: the fp32
    chain's ISTFT-head epilogue calls sincosf, whose large-argument reduction keeps a 320-byte table in scratch (no
    register spills; WT_PLAN_FLAG_FP32_GEMM plans only)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    table = kr.kernel_table()
    assert len(table) > 100, len(table)                       # the parser found the kernels
    names = " ".join(table)
    for must in ("gemm16s_kernel<128, 192, 4, 2, 3, 2, 1, 0", "resblock16_kernel<32, 128, 1, false, 4, 32>",
                 "resblock16_kernel<32, 128, 1, false, 2, 32>", "lstm_persist_kernel", "lstm_step_kernel<true>"):
        assert must in names, must
    allowed_scratch = ("gemm_kernel<128, 128, 2, 2, 0, 4>",)
    bad = []
    for name, t in table.items():
        if t.get("vgpr_spill_count", 0):
            bad.append((name, "vgpr_spill_count", t["vgpr_spill_count"]))
        if t.get("private_segment_fixed_size", 0) and not any(a in name for a in allowed_scratch):
            bad.append((name, "private_segment_fixed_size", t["private_segment_fixed_size"]))
    assert not bad, bad


def test_step_runner_lanes_are_single_rank_only():
    """lanes > 1 (steps in flight on separate HIP streams) is the single-rank pipeline: together with the sharded exchange the
    runner refuses it, so the N = 1 and N > 1 bench lines always measure the same schedule."""
    from wavtokenizer_amd.sharding import StepRunner
    wav = torch.zeros((2, 16))
    with pytest.raises(ValueError, match="lanes"):
        StepRunner(object(), wav, torch.tensor([0]), object(), 2, 0, gather=True, backend="gloo", lanes=2)
    r = StepRunner(object(), wav, torch.tensor([0]), None, 1, 0, gather=True, lanes=1)      # world 1: no exchange, one lane
    assert r.lanes == 1 and r.streams is None and not r.gather


def test_product_library_reads_no_environment_and_carries_no_lab_kernels():
    """the product library must not reach getenv from any launch path and must not carry the
    timing-experiment kernel builds or the LSTM fault hook.  Every environment switch of csrc/ goes through lab_env(), a
    constant in the product build (common.h): no object of the product build references getenv at all; the LAB build
    (make lab -> tools/lib/libwavtok_hip_lab.so, what tools/*.py and tests/lab load through WAVTOK_HIP_LIB) does."""
    import glob
    import subprocess
    csrc = os.path.join(ROOT, "wavtokenizer_amd", "csrc")
    objs = sorted(glob.glob(os.path.join(csrc, "build", "*.o")))
    assert len(objs) >= 10, objs
    for o in objs:
        undefined = subprocess.run(["nm", "-u", o], capture_output=True, text=True, check=True).stdout
        assert "getenv" not in undefined, o
    src = "".join(open(f).read() for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")))
    assert "getenv(" not in src                                  # only common.h's lab_env (LAB builds) calls it
    dyn = subprocess.run(["nm", "-D", os.path.join(ROOT, "wavtokenizer_amd", "libwavtok_hip.so")], capture_output=True, text=True, check=True).stdout
    assert " U getenv" not in dyn
    # no experiment instantiations among the product's kernels: gemm16s_kernel<..., DBG, ...> has DBG == 0 everywhere, the
    # resblock16 ablation builds (<..., true, ...>) and the LSTM phase-trace builds (<..., true>) are absent
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    names = list(kr.kernel_table())
    for n in names:
        m = re.search(r"gemm16s_kernel<(.*?)>", n)
        if m:
            args = [a.strip() for a in m.group(1).split(",")]
            assert args[7] == "0", n                             # template argument DBG
        m = re.search(r"resblock16_kernel<(.*?)>", n)
        if m:
            assert m.group(1).split(",")[3].strip() == "false", n
        m = re.search(r"lstm_persist_kernel<(.*?)>", n)
        if m:
            assert m.group(1).split(",")[1].strip() == "false", n
    assert not os.path.exists(os.path.join(ROOT, "wavtokenizer_amd", "libwavtok_hip_prev.so"))
    lab = os.path.join(ROOT, "tools", "lib", "libwavtok_hip_lab.so")
    if os.path.exists(lab):
        lab_dyn = subprocess.run(["nm", "-D", lab], capture_output=True, text=True, check=True).stdout
        assert " U getenv" in lab_dyn
        lab_names = list(kr.kernel_table(lab))
        assert len(lab_names) > len(names)


def test_check_codes_mode_defaults_to_sync_and_is_validated(monkeypatch):
    """codes_to_features must raise for the offending call by default (F.embedding does:
    decoder/pretrained.py:236); 'deferred' is opt-in, '1' means sync, unknown values are refused."""
    from wavtokenizer_amd import WavTokenizer, ARCH_HOP600
    monkeypatch.delenv("WAVTOK_CHECK_CODES", raising=False)
    assert WavTokenizer.from_arch(ARCH_HOP600)._check_codes == "sync"
    monkeypatch.setenv("WAVTOK_CHECK_CODES", "1")
    assert WavTokenizer.from_arch(ARCH_HOP600)._check_codes == "sync"
    monkeypatch.setenv("WAVTOK_CHECK_CODES", "deferred")
    assert WavTokenizer.from_arch(ARCH_HOP600)._check_codes == "deferred"
    monkeypatch.setenv("WAVTOK_CHECK_CODES", "0")
    assert WavTokenizer.from_arch(ARCH_HOP600)._check_codes == "off"
    monkeypatch.setenv("WAVTOK_CHECK_CODES", "later")
    with pytest.raises(ValueError, match="WAVTOK_CHECK_CODES"):
        WavTokenizer.from_arch(ARCH_HOP600)


def test_strict_status_is_automatic_for_small_batches():
    """VERDICT r03 #3c: calls of up to the graph batch limit synchronise, check and repeat a failed call (an infer.py-style
    caller never receives poisoned tensors); larger batches stay asynchronous; both can be forced."""
    from wavtokenizer_amd import WavTokenizer, ARCH_HOP600
    m = WavTokenizer.from_arch(ARCH_HOP600)
    assert m._is_strict(1) and m._is_strict(16) and not m._is_strict(17) and not m._is_strict(64)
    m.set_graph_max_clips(0)
    assert not m._is_strict(1)
    m.set_strict_status(True)
    assert m._is_strict(64)
    m.set_strict_status(False)
    assert not m._is_strict(1)
    m.set_strict_status(None)
    m.set_graph_max_clips(16)
    assert m._is_strict(4)


def test_range_sites_select_per_plan_kind():
    """An overflow answered in one ConvNeXt block must not re-plan the encoder (the site mask a plan kind sees)."""
    from wavtokenizer_amd import WavTokenizer, ARCH_HOP600, _capi
    from wavtokenizer_amd.pretrained import site_name
    m = WavTokenizer.from_arch(ARCH_HOP600)
    m._fp32_sites = (1 << (_capi.WT_SITE_CNX0 + 5)) | (1 << _capi.WT_SITE_HEAD)
    assert m._sites(_capi.WT_PLAN_ENCODE) == 0
    assert m._sites(_capi.WT_PLAN_DECODE) == m._fp32_sites
    assert m._sites(_capi.WT_PLAN_HEAD) == 1 << _capi.WT_SITE_HEAD
    m._fp32_sites |= 1 << _capi.WT_SITE_ENCODER
    assert m._sites(_capi.WT_PLAN_ENCODE) == 1 and not (m._sites(_capi.WT_PLAN_DECODE) & 1)
    assert site_name(_capi.WT_SITE_CNX0 + 5) == "backbone.convnext.5" and "head" in site_name(_capi.WT_SITE_HEAD)


def test_plan_cache_bounds_streams():
    """plans are per stream; a caller that uses a new stream per request must not fill the cache with plans of
    streams it never uses again (host logic of _Engine, no GPU: keys only)."""
    from wavtokenizer_amd.pretrained import _Engine
    e = _Engine()
    e.max_streams = 2
    dropped = []
    e.drop = lambda pred: dropped.append([k for k in list(e.plans) if pred(k)]) or [e.plans.pop(k) for k in list(e.plans) if pred(k)]
    for sp in (11, 22, 33):
        e.plans[(0, 1, 100, 0, sp)] = (None, None)
        e._touch_stream(sp)
    assert e.stream_lru == [22, 33] and (0, 1, 100, 0, 11) not in e.plans and (0, 1, 100, 0, 33) in e.plans
    e._touch_stream(22)
    assert e.stream_lru == [33, 22]
    e.plans.clear()
