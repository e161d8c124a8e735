"""CPU-only checks: the C-ABI library loads and exports every symbol include/vv_hip.h declares (no compute calls),
the host-side schedule/processor/streamer logic, the checkpoint layout accounting, the N>1 sharding path over gloo,
and that the product refuses to run without its HIP extension / a GPU (no CPU fallback)."""
import os
import re
import sys
import threading

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
    assert lib.vv_abi_version() == 2


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


def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
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
