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


def split_frames(rows):
    cuts = [i for i, r in enumerate(rows) if "advance_lens" in r["Kernel_Name"] or "llm_tail" in r["Kernel_Name"]]
    frames = [rows[cuts[i] + 1: cuts[i + 1] + 1] for i in range(len(cuts) - 1)]
    lens = collections.Counter(len(f) for f in frames)
    n_typ = lens.most_common(1)[0][0]
    return [f for f in frames if len(f) == n_typ], n_typ


def label_ops(frame, model="1.5b"):
    """(region, op, m, n, k) for every kernel of a steady-state frame: the launch sequence of a frame is fixed, so the position inside
    its region identifies the logical op (and with it the GEMV shape) even where two shapes share one template instantiation and grid."""
    sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
    from vibevoice_rocm_amd.config import VVConfig
    cfg = VVConfig.preset(model)
    H, I, D, F = cfg.hidden, cfg.inter, cfg.head_hidden, cfg.head_ffn
    qkv_n = (cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim
    out, region, idx, gathers = [], "head_pre", 0, 0
    for r in frame:
        n = short(r["Kernel_Name"])
        if "head_init_kernel" in n:
            region, idx = "head", -1
        elif "conv_ctx_gather" in n:
            gathers += 1
            region, idx = ("decoder" if gathers == 1 else "semantic"), -1
        elif "rope_table_kernel" in n:
            region, idx = "llm", -1
        op, shp = n.split("<")[0], (0, 0, 0)
        if region == "head" and idx >= 0:
            j = idx % (2 * cfg.head_layers + 1)
            if j == 2 * cfg.head_layers:
                op, shp = "head.boundary(G=[PF;F], fp32)", (1, D + cfg.latent, D)
            elif j % 2 == 0:
                op, shp = "head.gate_up", (2, F, D)
            else:
                op, shp = "head.down", (2, D, F)
        elif region == "llm" and idx >= 0 and "llm_tail" not in n:
            j = idx % 5
            op, shp = [("llm.qkv", (2, qkv_n, H)), ("llm.attn_decode", (0, 0, 0)), ("llm.o", (2, H, cfg.q_dim)), ("llm.gate_up", (2, I, H)), ("llm.down", (2, H, I))][j]
        out.append((region, op, shp))
        idx += 1
        if "adaln_" in n and region == "head_pre":      # the solver loop starts behind the adaLN kernel (head_init is part of head_pre_kernel since round 3)
            region, idx = "head", 0
        if "conv_ctx_scatter" in n and gathers == 2:
            region, idx = "connect", 0
    return out


def reduce(path, out_prefix=None, model="1.5b"):
    rows = list(csv.DictReader(open(path)))
    is_pmc = "Counter_Value" in rows[0]
    if is_pmc:
        rows = [r for r in rows if r.get("Counter_Name") == "FETCH_SIZE"]
        for r in rows:
            r.setdefault("Grid_Size_X", r.get("Grid_Size", "0")); r.setdefault("Grid_Size_Y", "1"); r.setdefault("Workgroup_Size_X", r.get("Workgroup_Size", "1"))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    typ, n_typ = split_frames(rows)
    labels = label_ops(typ[0], model)
    durs = sorted((int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in typ)
    med = typ[len(typ) // 2]
    lines = [f"{len(typ)} steady-state frames of {n_typ} kernels; frame time min {durs[0]:.1f} / median {durs[len(durs) // 2]:.1f} / max {durs[-1]:.1f} us"]
    t0 = int(med[0]["Start_Timestamp"])
    prev = t0
    for r, (reg, op, shp) in zip(med, labels):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        lines.append(f"{(s - t0) / 1e3:8.1f}  gap {(s - prev) / 1e3:5.1f}  dur {(e - s) / 1e3:6.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):<3d} wg {r['Workgroup_Size_X']:>4}  {reg:9s} {op:34s} {short(r['Kernel_Name'])}")
        prev = e
    # per logical op (region, op, kernel, grid): launches per frame, average in-graph duration, and FETCH_SIZE when this is a counter pass
    stat = collections.OrderedDict()
    for f in typ:
        for r, (reg, op, shp) in zip(f, labels):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            key = (reg, op, short(r["Kernel_Name"]), f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}", r["Workgroup_Size_X"], shp)
            st = stat.setdefault(key, [0, 0.0, 0.0])
            st[0] += 1
            st[1] += (e - s) / 1e3
            if is_pmc:
                st[2] += float(r["Counter_Value"])
    tot = sum(v[1] for v in stat.values()) / len(typ)
    agg = [f"per-frame kernel time {tot:.1f} us; by (region, op, kernel, grid):"]
    csv_rows = [["region", "op", "kernel", "grid", "workgroup", "m", "n", "k", "launches_per_frame", "avg_us", "per_frame_us", "share", "weight_bytes", "GBps", "fetch_bytes_per_launch"]]
    for k, v in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        reg, op, kern, grid, wg, (m, n, kk) = k
        avg = v[1] / v[0]
        wbytes = (4 if "fp32" in op else 2) * n * kk * (2 if op.endswith("gate_up") else 1)
        gbs = wbytes / avg / 1e3 if wbytes else 0.0
        fetch = 2.0 * 1024.0 * v[2] / v[0] if is_pmc else ""
        agg.append(f"{v[0] / len(typ):7.1f}/frame  dur {avg:7.2f} us  per-frame {v[1] / len(typ):7.1f} us {v[1] / len(typ) / tot * 100:5.1f}%  grid {grid:>9} wg {wg:>4}  {reg:9s} {op:34s} {kern}"
                   + (f"  {gbs:6.0f} GB/s" if wbytes else "") + (f"  fetch {fetch / 1e6:7.2f} MB" if is_pmc else ""))
        csv_rows.append([reg, op, kern, grid, wg, m, n, kk, round(v[0] / len(typ), 2), round(avg, 3), round(v[1] / len(typ), 2), round(v[1] / len(typ) / tot, 4), wbytes, round(gbs, 1), fetch])
    text = "\n".join(lines) + "\n\n" + "\n".join(agg) + "\n"
    if out_prefix:
        if not is_pmc:
            open(out_prefix + "_frame_timeline.txt", "w").write(text)
        with open(out_prefix + ("_pmc_per_shape.csv" if is_pmc else "_per_shape.csv"), "w", newline="") as f:
            csv.writer(f).writerows(csv_rows)
        if not is_pmc:       # ties the table to the kernel source it was measured on (bench.py reads the in-graph durations only when it matches)
            sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
            import bench as _b
            open(out_prefix + "_per_shape.sha", "w").write(_b.kernel_source_sha() + "\n")
        if is_pmc:      # profiles/pmc_traffic.json: HBM-side bytes per launch of the GEMV shapes, tied to the kernel source they were measured on
            import json
            sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
            import bench
            per = {}
            for row in csv_rows[1:]:
                if row[5] and row[14] != "":
                    per[f"{row[5]}x{row[6]}x{row[7]}"] = int(row[14])
            json.dump({"kernel_src_sha": bench.kernel_source_sha(), "per_launch_bytes": per,
                       "note": "rocprofv3 --pmc FETCH_SIZE --kernel-trace (own pass) over hipGraph replays of real generate() frames (tools/frame_trace.py run): "
                               "FETCH_SIZE is in KB and on gfx950 counts 64 B per 128-B request of a wide coalesced stream, so bytes = 2 x FETCH_SIZE x 1024 "
                               "(MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are counted (memory-side requests of L2), so the head's "
                               "cache-resident matrices still show their full size"}, open(out_prefix + "_pmc_traffic.json", "w"), indent=1, sort_keys=True)
    print(text)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2:])
    else:
        reduce(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else "1.5b")
