"""Where a row-batched step spends its time: graph A (8-row LLM step + tail), graph H (8-row diffusion sampling) and the four conv tails
(one hipGraph per dialogue on its own stream) replayed in isolation after a short generate() has set every buffer up."""
import sys, time, types
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
import bench
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 2024, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
args = types.SimpleNamespace(frames=40, voice_frames=203, cfg_scale=2.0)
import os
if os.environ.get("VV_GEMV_OPT"):
    m.engine.lib.vv_tune(b"gemv_opt", int(os.environ["VV_GEMV_OPT"]))
bench.batched_leg(m, cfg, args, B, row_batch=True)
rb = m._rowbatch[(B, 0)]
lib = rb.lib
torch.cuda.synchronize()
rb.set_active(0, True)
for b in range(B):
    rb.set_active(b, False)          # positions stay put while the graphs are replayed


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


gA = [g for k, g in rb._graphs.items() if k[0] == "A"][0]
gH = [g for k, g in rb._graphs.items() if k[0] == "H"][0]
print(f"graph A (LLM step, {2 * B} rows + tail): {timeit(lambda: lib.vv_graph_launch(gA, rb.sp)):.3f} ms", flush=True)
print(f"graph H (diffusion sampling, {2 * B} rows): {timeit(lambda: lib.vv_graph_launch(gH, rb.sp)):.3f} ms", flush=True)
conv = []
for b in range(B):
    e = rb.lanes[b]
    conv.append((e, [g for k, g in e._graphs.items() if k[0] == "RBconv"][0]))
print(f"conv tail, one dialogue alone: {timeit(lambda: lib.vv_graph_launch(conv[0][1], conv[0][0].sp)):.3f} ms", flush=True)


def all_conv():
    for e, g in conv:
        lib.vv_graph_launch(g, e.sp)


print(f"conv tails, {B} dialogues on {B} streams: {timeit(all_conv):.3f} ms", flush=True)


def all_conv_latency():
    # as a step runs them: all tails start together, the next round waits for all of them
    torch.cuda.synchronize()
    for e, g in conv:
        lib.vv_graph_launch(g, e.sp)
    torch.cuda.synchronize()


print(f"conv tails, {B} dialogues started together, waited for (latency, incl. ~2 host syncs): {timeit(all_conv_latency, 30):.3f} ms", flush=True)


def host_cost(fn, n=3):
    """host time of an enqueue (the call returns before the GPU runs the graph): few launches, so the queue never fills"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return dt


print(f"host time to enqueue: graph A {host_cost(lambda: lib.vv_graph_launch(gA, rb.sp)):.3f} ms, graph H {host_cost(lambda: lib.vv_graph_launch(gH, rb.sp)):.3f} ms, "
      f"one conv tail {host_cost(lambda: lib.vv_graph_launch(conv[1][1], conv[1][0].sp)):.3f} ms", flush=True)
e0 = m.engine
g1 = {k[0]: g for k, g in e0._graphs.items()}
print("single-dialogue graphs on the main engine:", {k: round(timeit(lambda g=g: lib.vv_graph_launch(g, e0.sp)), 3) for k, g in g1.items() if k in ("A", "B")})
