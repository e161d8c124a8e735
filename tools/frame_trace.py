"""Steady-state frames of generate() under `rocprofv3 --kernel-trace`, and the reducer that turns the CSV into a per-kernel timeline of
one frame plus per-(kernel, grid) statistics.

  run:     rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft -- python3 tools/frame_trace.py run [--model 1.5b] [--frames 40]
  reduce:  python tools/frame_trace.py reduce <kernel_trace.csv> [out_prefix]

A "frame" is everything between two llm_tail (token + bookkeeping) kernels (diffusion tail of frame i, then the LLM step that picks token i+1)."""
import csv
import collections
import sys


def run(argv):
    import argparse
    import torch
    sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="1.5b")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--prompt", type=int, default=330)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args(argv)
    cfg = VVConfig.preset(a.model)
    sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    m.set_ddpm_inference_steps(a.steps)
    V = cfg.vocab

    class Tok:
        speech_start_id, speech_end_id, speech_diffusion_id, eos_token_id, bos_token_id, pad_id = V - 4, V - 3, V - 2, V - 1, None, 0
    g = torch.Generator().manual_seed(1)
    ids = torch.cat([torch.randint(0, 1000, (a.prompt - 1,), generator=g), torch.tensor([V - 4])])
    forced = [V - 2] * a.frames + [V - 3, V - 1]
    noise = torch.randn(a.frames, cfg.latent, generator=g)
    for _ in range(2):
        out = m.generate(input_ids=ids[None], tokenizer=Tok(), cfg_scale=2.0, forced_tokens=forced, noise=noise)
    print("frames", out.speech_outputs[0].shape[-1] // cfg.hop)


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:64]


def reduce(path, out_prefix=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cuts = [i for i, r in enumerate(rows) if "advance_lens" in r["Kernel_Name"] or "llm_tail_kernel" in r["Kernel_Name"]]
    frames = [rows[cuts[i] + 1: cuts[i + 1] + 1] for i in range(len(cuts) - 1)]
    lens = collections.Counter(len(f) for f in frames)
    n_typ = lens.most_common(1)[0][0]
    typ = [f for f in frames if len(f) == n_typ]
    durs = sorted((int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in typ)
    med = typ[len(typ) // 2]
    lines = [f"{len(typ)} steady-state frames of {n_typ} kernels; frame time min {durs[0]:.1f} / median {durs[len(durs) // 2]:.1f} / max {durs[-1]:.1f} us"]
    t0 = int(med[0]["Start_Timestamp"])
    prev = t0
    for r in med:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        lines.append(f"{(s - t0) / 1e3:8.1f}  gap {(s - prev) / 1e3:5.1f}  dur {(e - s) / 1e3:6.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):<3d} wg {r['Workgroup_Size_X']:>4}  {short(r['Kernel_Name'])}")
        prev = e
    # per-position statistics over all steady-state frames (the launch sequence of a frame is fixed, so position = logical op)
    stat = collections.OrderedDict()
    for f in typ:
        prev = None
        for pos, r in enumerate(f):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            key = (short(r["Kernel_Name"]), f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}", r["Workgroup_Size_X"])
            st = stat.setdefault(key, [0, 0.0, 0.0])
            st[0] += 1
            st[1] += (e - s) / 1e3
            if prev is not None:
                st[2] += max(0.0, (s - prev) / 1e3)
            prev = e
    tot = sum(v[1] + v[2] for v in stat.values()) / len(typ)
    agg = [f"per-frame kernel+gap time {tot:.1f} us; by (kernel, grid, workgroup):"]
    for k, v in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        agg.append(f"{v[0] / len(typ):7.1f}/frame  dur {v[1] / v[0]:7.2f} us  gap {v[2] / v[0]:5.2f}  per-frame {(v[1] + v[2]) / len(typ):7.1f} us {(v[1] + v[2]) / len(typ) / tot * 100:5.1f}%  grid {k[1]:>9} wg {k[2]:>4}  {k[0]}")
    text = "\n".join(lines) + "\n\n" + "\n".join(agg) + "\n"
    if out_prefix:
        open(out_prefix + "_frame_timeline.txt", "w").write(text)
    print(text)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2:])
    else:
        reduce(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
