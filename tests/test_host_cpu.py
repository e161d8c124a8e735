"""CPU-only checks: the C-ABI library loads and exports every symbol include/vv_hip.h declares (no compute calls),
the host-side schedule/processor/streamer logic, the checkpoint layout accounting, the N>1 sharding path over gloo,
and that the product refuses to run without its HIP extension / a GPU (no CPU fallback)."""
import os
import re
import sys
import threading
import time

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from vibevoice_rocm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vv_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vv_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = _lib.load()                       # also verifies every struct size against the ctypes mirror
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.vv_abi_version() == 6


def test_integration_md_struct_mirrors_match_the_library():
    """The ctypes mirrors INTEGRATION.md shows a reference maintainer are executed and checked against the built library
    (vv_sizeof) and against the product's own mirrors (field names, order, sizes): a stale snippet misreads device pointers."""
    import ctypes as C
    from vibevoice_rocm_amd import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", md, flags=re.S) if b.lstrip().startswith("# abi-mirror")]
    assert len(blocks) == 1
    ns = {}
    exec(blocks[0], ns)
    lib = _lib.load()
    found = {k: v for k, v in ns.items() if isinstance(v, type) and issubclass(v, C.Structure) and k.startswith("vv_")}
    assert {"vv_head_layer", "vv_head", "vv_dpm_coef"} <= set(found)
    mirrors = dict(_lib._STRUCTS, vv_w8=_lib.W8)
    for name, cls in found.items():
        if name != "vv_w8":
            assert lib.vv_sizeof(name.encode()) == C.sizeof(cls), (name, lib.vv_sizeof(name.encode()), C.sizeof(cls))
        ours = mirrors[name]
        assert [f[0] for f in cls._fields_] == [f[0] for f in ours._fields_], name
        assert [C.sizeof(f[1]) for f in cls._fields_] == [C.sizeof(f[1]) for f in ours._fields_], name
    assert f"ABI v{lib.vv_abi_version()}" in blocks[0]


def test_checkpoint_dir_roundtrip_reference_layout(tmp_path):
    """save_checkpoint_dir writes the reference's layout (config.json schema of vibevoice/configs/*.json incl. the vibepod_* model_type
    keys, preprocessor_config.json, safetensors shards + index as scripts/convert_nnscaler_checkpoint_to_transformers.py:92-123 does);
    VVConfig.from_pretrained / load_state_dict_from_dir / VibeVoiceProcessor.from_pretrained read it back (CPU part of the drop-in
    path; the GPU part is tests/test_hip_configs.py::test_drop_in_checkpoint_dir_through_reference_import_paths)."""
    import dataclasses
    import json
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import load_state_dict_from_dir, save_checkpoint_dir
    from vibevoice_rocm_amd.processor import SyntheticTokenizer, VibeVoiceProcessor
    from vibevoice_rocm_amd.synth import synth_state_dict
    for tie in (True, False):
        cfg = dataclasses.replace(VVConfig.preset("tiny"), tie=tie)
        sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 3).items()}
        path = os.path.join(tmp_path, f"ck{int(tie)}")
        save_checkpoint_dir(path, cfg, sd, max_shard_bytes=40_000)
        assert VVConfig.from_pretrained(path) == cfg
        back = load_state_dict_from_dir(path)
        assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
        assert ("lm_head.weight" in back) == (not tie)
        assert len([f for f in os.listdir(path) if f.endswith(".safetensors")]) > 2
        p = VibeVoiceProcessor.from_pretrained(path, tokenizer=SyntheticTokenizer(cfg.vocab))
        assert p.speech_tok_compress_ratio == cfg.hop and p.db_normalize
        with pytest.raises(OSError):
            VibeVoiceProcessor.from_pretrained(path)                  # no tokenizer files, no hub access
    j = json.load(open(os.path.join(path, "config.json")))
    ref_json = "/root/reference/vibevoice/configs/qwen2.5_1.5b_64k.json"
    if os.path.exists(ref_json):                                       # build container only: same key sets as the reference's own JSON
        r = json.load(open(ref_json))
        for sec in ("acoustic_tokenizer_config", "semantic_tokenizer_config", "diffusion_head_config"):
            assert set(r[sec]) == set(j[sec]), (sec, set(r[sec]) ^ set(j[sec]))
        assert set(r["decoder_config"]) <= set(j["decoder_config"]) | {"torch_dtype"}
        assert VVConfig.from_json_dict(r) == VVConfig.preset("1.5b")
        r7 = json.load(open(ref_json.replace("1.5b_64k", "7b_32k")))
        assert VVConfig.from_json_dict(r7) == VVConfig.preset("7b")


def test_no_cpu_fallback():
    from vibevoice_rocm_amd import _lib
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.engine import Engine
    with pytest.raises(_lib.VVError):
        Engine(VVConfig.preset("tiny"), {}, device="cpu")
    src = "".join(open(os.path.join(ROOT, "vibevoice_rocm_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "vibevoice_rocm_amd")) if f.endswith(".py"))
    assert "oracle" not in src.replace("oracle/", ""), "the product must never import the oracle"


def test_schedule_matches_reference_tables():
    from vibevoice_rocm_amd.schedule import DPMSolverMultistepScheduler, timestep_sinusoid
    g = load_golden("scheduler")
    s = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_schedule="cosine", prediction_type="v_prediction")
    np.testing.assert_allclose(s.alphas_cumprod.numpy(), g["alphas_cumprod"], rtol=1e-6)
    for n in (10, 20, 50):
        s.set_timesteps(n)
        assert s.timesteps.tolist() == g[f"timesteps_{n}"].tolist()
        np.testing.assert_allclose(s.sigmas.numpy(), g[f"sigmas_{n}"], rtol=5e-7)
        assert [c["order"] for c in s.coefs] == [1] + [2] * (n - 2) + [1]
    # trajectory with v = 0.1 x through the host coefficients (same update the vv_dpm_step kernel applies)
    s.set_timesteps(20)
    x = torch.from_numpy(g["traj_x0_20"]).clone()
    m_prev = None
    for i, c in enumerate(s.coefs):
        x0 = c["alpha_s"] * x - c["sigma_s"] * (0.1 * x)
        xn = c["cx"] * x - c["cd"] * x0
        if c["order"] == 2:
            xn = xn - 0.5 * c["cd"] * (c["rinv"] * (x0 - m_prev))
        x, m_prev = xn, x0
        np.testing.assert_allclose(x.numpy(), g["traj_20"][i], rtol=2e-4, atol=2e-5)
    # the SDE solver main.py selects is built (noise coefficient > 0 except on the final sigma-0 step); other variants are refused
    sde = DPMSolverMultistepScheduler.from_config(s.config, algorithm_type="sde-dpmsolver++", beta_schedule="squaredcos_cap_v2")
    sde.set_timesteps(20)
    assert all(c["cn"] > 0 for c in sde.coefs[:-1]) and sde.coefs[-1]["cn"] == 0.0 and sde.coefs[-1]["cx"] == 0.0
    assert all(c["cn"] == 0.0 for c in s.coefs)
    with pytest.raises(NotImplementedError):
        DPMSolverMultistepScheduler.from_config(s.config, algorithm_type="dpmsolver")
    # bf16 quirk of the reference's bf16 run: timesteps are rounded to bf16 before the sinusoid
    assert torch.tensor([949.0]).bfloat16().float().item() == 948.0
    e = timestep_sinusoid([949], 256, bf16_quirk=True)
    assert torch.allclose(e, timestep_sinusoid([948], 256).bfloat16().float())


def test_checkpoint_layout_accounting():
    """Parameter counts of the 1.5B layout reproduce SURVEY.md §8d's table (so bytes/frame in bench.py is the survey's)."""
    sys.path.insert(0, ROOT)
    import bench
    from vibevoice_rocm_amd.config import VVConfig
    cfg = VVConfig.preset("1.5b")
    b, b_res, tot = bench.bytes_per_frame(cfg, 20, 0, 2)
    assert tot["llm"] == 1_310_340_608 and tot["head"] == 123_279_360
    assert tot["dec"] == 343_695_969 and tot["sem"] == 344_613_600 and tot["conn"] == 5_022_720
    assert abs(b / 1e9 - 8.939) < 0.002
    cfg7 = VVConfig.preset("7b")
    _, _, tot7 = bench.bytes_per_frame(cfg7, 20, 0, 2)
    assert tot7["llm"] == 6_525_621_760 and tot7["head"] == 669_333_504 and tot7["conn"] == 26_399_744


def test_processor_prompt_and_audio(tmp_path):
    from vibevoice_rocm_amd.processor import VibeVoiceProcessor, SyntheticTokenizer, AudioNormalizer, load_wav
    tok = SyntheticTokenizer(1024)
    p = VibeVoiceProcessor(tokenizer=tok)
    voice = (0.1 * np.random.RandomState(0).randn(3200 * 2 + 100)).astype(np.float32)
    out = p(text=["Speaker 1: Hello there.\nSpeaker 2: Hi!", "Speaker 1: Short."], voice_samples=[[voice, voice[:4000]], [voice[:3300]]],
            padding=True, return_tensors="pt", return_attention_mask=True)
    ids, am, sm = out["input_ids"], out["attention_mask"], out["speech_input_mask"]
    assert ids.shape == am.shape == sm.shape and ids.shape[0] == 2
    assert am[1, 0] == 0 and am[1, -1] == 1 and ids[1, 0] == tok.pad_id           # LEFT padding
    assert ids[0, -1] == tok.speech_start_id and ids[1, -1] == tok.speech_start_id
    assert int(sm[0].sum()) == 3 + 2 and int(sm[1].sum()) == 2                       # ceil(len / 3200) placeholders per voice
    assert (ids[sm] == tok.speech_diffusion_id).all()
    assert out["speech_tensors"].shape == (3, 6500) and out["speech_masks"].sum().item() == 7
    assert out["parsed_scripts"][0] == [(0, " Hello there."), (1, " Hi!")]           # ids normalised to start at 0
    x = AudioNormalizer()(voice)
    assert abs(20 * np.log10(np.sqrt(np.mean(x ** 2))) + 25) < 0.05
    path = os.path.join(tmp_path, "o", "a.wav")
    p.save_audio(torch.from_numpy(x)[None], output_path=path)
    y = load_wav(path)
    assert y.shape == x.shape and np.max(np.abs(y - x)) < 1e-4 + 1 / 32768


def test_processor_vs_reference_fixture(tmp_path):
    """ids / masks / padded speech tensors bit-exact against the reference's own VibeVoiceProcessor.__call__ under the same stand-in
    tokenizer (tests/golden/processor.npz, oracle/gen/make_golden.py::gen_processor; reference
    vibevoice/processor/vibevoice_processor.py:148-229,231-289,291-389,391-444,446-494,496-616): 2- and 4-speaker scripts, a left-padded
    batch of both, no voices, a .txt file (plain lines -> Speaker 1, empty texts dropped) and a .json file (malformed entries skipped)."""
    from vibevoice_rocm_amd.processor import VibeVoiceProcessor, SyntheticTokenizer
    g = load_golden("processor")
    rng = np.random.Generator(np.random.PCG64(int(g["seed"])))
    voices = [(a * rng.standard_normal(int(n))).astype(np.float32) for a, n in zip(g["amps"], g["lens"])]
    s2, s4 = str(g["script2"]), str(g["script4"])
    p = VibeVoiceProcessor(tokenizer=SyntheticTokenizer(1024))
    kw = dict(padding=True, return_tensors="pt", return_attention_mask=True)
    txt, js = os.path.join(tmp_path, "script.txt"), os.path.join(tmp_path, "script.json")
    with open(txt, "w", encoding="utf-8") as f:
        f.write(str(g["txt"]))
    with open(js, "w", encoding="utf-8") as f:
        f.write(str(g["json"]))
    cases = dict(two=p(text=[s2], voice_samples=[voices[:2]], **kw), four=p(text=[s4], voice_samples=[voices], **kw),
                 batch=p(text=[s2, s4], voice_samples=[voices[:2], voices], **kw), novoice=p(text=s2, **kw),
                 txtfile=p(text=[txt], voice_samples=[voices[:3]], **kw), jsonfile=p(text=[js], voice_samples=[voices[:2]], **kw))
    for tag, out in cases.items():
        for k in ("input_ids", "attention_mask", "speech_input_mask"):
            assert out[k].dtype == {"input_ids": torch.long, "attention_mask": torch.long, "speech_input_mask": torch.bool}[k]
            assert np.array_equal(out[k].numpy(), g[f"{tag}_{k}"]), (tag, k)
        if f"{tag}_speech_tensors" in g:
            assert out["speech_tensors"].dtype == torch.float32 and out["speech_masks"].dtype == torch.bool
            assert np.array_equal(out["speech_tensors"].numpy(), g[f"{tag}_speech_tensors"]), tag
            assert np.array_equal(out["speech_masks"].numpy(), g[f"{tag}_speech_masks"]), tag
        else:
            assert out["speech_tensors"] is None and out["speech_masks"] is None
        assert repr(out["parsed_scripts"]) == str(g[f"{tag}_parsed"]), tag
        assert repr(out["all_speakers_list"]) == str(g[f"{tag}_speakers"]), tag
    assert float(np.abs(g["four_speech_tensors"][2]).max()) <= 1.0 and voices[2].std() > 1.0      # the anti-clip branch was exercised


def test_streamer_threaded():
    from vibevoice_rocm_amd.streamer import AudioStreamer
    st = AudioStreamer(batch_size=2, stop_signal=None, timeout=5)
    got = []

    def consume():
        for chunk in st.get_stream(0):
            got.append(chunk)

    th = threading.Thread(target=consume)
    th.start()
    for i in range(3):
        st.put(torch.full((1, 1, 4), float(i)), torch.tensor([0]))
    st.end(torch.tensor([0]))
    th.join(5)
    assert len(got) == 3 and got[2].shape == (1, 4) and st.finished_flags == [True, False]
    st.put(torch.zeros(1, 1, 4), torch.tensor([0]))        # ignored after end()
    assert st.audio_queues[0].empty()
    st.end()
    assert st.finished_flags == [True, True]


def test_streamer_batch_iterator_threaded():
    """`for d in streamer`: {sample: chunk} dicts until every sample has ended (reference streamer.py:107-147), producer on another thread."""
    from vibevoice_rocm_amd.streamer import AudioStreamer
    st = AudioStreamer(batch_size=3, timeout=5)

    def produce():
        for i in range(4):
            st.put(torch.full((2, 1, 4), float(i)), torch.tensor([0, 2]))
            if i == 1:
                st.end(torch.tensor([2]))
        st.end()

    th = threading.Thread(target=produce)
    th.start()
    seen = {0: [], 1: [], 2: []}
    for d in st:
        for k, v in d.items():
            seen[k].append(float(v.mean()))
    th.join(5)
    assert seen == {0: [0.0, 1.0, 2.0, 3.0], 1: [], 2: [0.0, 1.0]}
    with pytest.raises(ValueError):
        st.get_stream(3)


def test_async_streamer_from_generating_thread():
    """AsyncAudioStreamer (reference streamer.py:150-264): queues live on the consumer's event loop, put()/end() come from the
    generating thread; per-sample `async for` and the batch `async for` both see every chunk in order, nothing after end()."""
    import asyncio
    from vibevoice_rocm_amd.streamer import AsyncAudioStreamer

    async def main():
        st = AsyncAudioStreamer(batch_size=2, timeout=5)

        def produce():
            for i in range(5):
                st.put(torch.full((2, 1, 8), float(i)), torch.tensor([0, 1]))
                time.sleep(0.002)
            st.end(torch.tensor([1]))
            st.put(torch.full((2, 1, 8), 9.0), torch.tensor([0, 1]))      # sample 1 has ended: only sample 0 receives this
            st.end()

        th = threading.Thread(target=produce)
        th.start()

        async def one(i):
            return [float(c.mean()) async for c in st.get_stream(i)]
        a, b = await asyncio.gather(one(0), one(1))
        th.join(5)
        assert a == [0.0, 1.0, 2.0, 3.0, 4.0, 9.0] and b == [0.0, 1.0, 2.0, 3.0, 4.0]
        assert st.finished_flags == [True, True]

        st2 = AsyncAudioStreamer(batch_size=3, timeout=5)

        def produce2():
            for i in range(6):
                st2.put(torch.full((1, 1, 8), float(i)), torch.tensor([i % 3]))
            st2.end()
        th2 = threading.Thread(target=produce2)
        th2.start()
        got = {0: [], 1: [], 2: []}
        async for d in st2:
            for k, v in d.items():
                got[k].append(float(v.mean()))
        th2.join(5)
        assert got == {0: [0.0, 3.0], 1: [1.0, 4.0], 2: [2.0, 5.0]}
        with pytest.raises(ValueError):
            async for _ in st2.get_stream(5):
                pass
    asyncio.run(main())


def _dist_worker(rank, world, port, q, bcast="broadcast"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["VV_BCAST"] = bcast
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vibevoice_rocm_amd import distributed as vd
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.synth import synth_state_dict
    cfg = VVConfig.preset("tiny")
    ref = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 7).items()}
    sd = vd.broadcast_state_dict(ref if rank == 0 else None, cfg, torch.float32, "cpu", src=0)
    ok = set(sd) == set(ref) and all(torch.equal(sd[k], ref[k]) for k in ref)
    items = vd.shard_items(5, rank, world)
    wav = torch.full((1, 100 * (rank + 1)), float(rank + 1))
    got = vd.gather_waveforms(wav, dst=0)
    if rank == 0:
        ok = ok and len(got) == world and all(g.shape == (1, 100 * (r + 1)) and float(g.mean()) == r + 1 for r, g in enumerate(got))
    else:
        ok = ok and got is None
    q.put((rank, ok, items))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True, [0, 2, 4]), (1, True, [1, 3])]


def test_sharding_world3_scatter_allgather_gloo():
    """The scatter + all-gather form of the checkpoint broadcast (VV_BCAST=scatter_allgather: the root sends 1/N of each blob to every
    peer, the peers exchange the pieces) on three gloo ranks: every rank ends with the exact state dict; ragged waveform gather; 7
    dialogues sharded 3 / 2 / 2."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dist_worker, args=(r, 3, port, q, "scatter_allgather")) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True), (2, True)]
    assert [it for _, _, it in res] == [[0, 3], [1, 4], [2]]


def test_hw_queue_default_is_set_on_import():
    """The package sets GPU_MAX_HW_QUEUES=8 unless the user chose a value (4 lanes + the copy stream oversubscribe the default 4)."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", "import os, vibevoice_rocm_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"], env=env, capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "8", out.stderr[-400:]
    env["GPU_MAX_HW_QUEUES"] = "4"
    out = subprocess.run([sys.executable, "-c", "import os, vibevoice_rocm_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"], env=env, capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "4"


def _run_bench(env_extra, *argv, timeout=240):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout, cwd=root)


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` with no torchrun around it starts two fresh rank processes (gloo here, stub body: the launcher,
    rendezvous and relay are the thing under test), relays rank 0's single JSON line and returns 0; a failing rank makes the parent
    fail and suppresses the result line; WORLD_SIZE != --gpus is an error in every case."""
    import json
    r = _run_bench({"VV_BENCH_STUB": "1", "VV_DIST_BACKEND": "gloo"}, "--gpus", "2")
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl"] == {"ranks": 2, "backend": "gloo"} and j["max_rank_plus_1"] == 2.0
    r = _run_bench({"VV_BENCH_STUB": "1", "VV_DIST_BACKEND": "gloo", "VV_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2")
    assert r.returncode == 3 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], (r.returncode, r.stdout)
    r = _run_bench({"VV_BENCH_STUB": "1", "WORLD_SIZE": "1", "RANK": "0"}, "--gpus", "2")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    r = _run_bench({"VV_BENCH_STUB": "1", "WORLD_SIZE": "2", "RANK": "0"}, "--gpus", "1")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_under_torchrun_two_ranks():
    """The driver's own launch line (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2) still works: the
    ranks take RANK / WORLD_SIZE from torchrun and nobody self-launches."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(VV_BENCH_STUB="1", VV_DIST_BACKEND="gloo")
    port = 31500 + os.getpid() % 1000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=240, cwd=root)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2, r.stdout
