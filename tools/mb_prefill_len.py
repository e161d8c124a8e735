"""LLM prompt prefill time by prompt length (HIP events on the engine stream, median of 5) - the first-chunk leg that scales with the script."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.engine import Engine
from vibevoice_rocm_amd.synth import synth_state_dict_torch
model = sys.argv[1] if len(sys.argv) > 1 else "1.5b"
cfg = VVConfig.preset(model)
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
eng = Engine(cfg, sd, device="cuda:0", dtype=torch.bfloat16)
V = cfg.vocab
for L0, chunk in ((330, 1024), (1040, 1024), (1500, 1024), (1500, 512)):
    eng.begin_sequence(2048, [V - 4, V - 3, V - 2, V - 1])
    x0 = torch.randn(L0, cfg.hidden, device="cuda") * 0.02
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(eng.stream):
            e0.record(eng.stream); eng.prefill(x0, row=0, pos0=0, chunk=chunk); e1.record(eng.stream)
        eng.stream.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[1:])
    print(f"{model} prefill L0={L0:5d} chunk={chunk:5d}: {ts[len(ts) // 2]:7.3f} ms", flush=True)
