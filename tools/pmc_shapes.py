"""HBM traffic per launch of the dominant weight-streaming GEMVs, by shape.
  step 1 (on the GPU box, own pass):  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc -- python3 tools/pmc_shapes.py run
  step 2:                             python tools/pmc_shapes.py reduce /tmp/pmc/*/*counter_collection.csv profiles/pmc_traffic.json
Every shape is launched N times on fresh weight copies (no cache reuse between launches); step 2 takes the gemv_stream dispatches in
launch order, N per shape.  FETCH_SIZE is in KB and on gfx950 counts 64 B per 128-B request for wide coalesced loads:
bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section)."""
import sys, json
SHAPES = [(2, 4608, 1536, True), (2, 8960, 1536, True), (2, 1536, 4608, False), (2, 1536, 8960, False), (2, 2048, 1536, False), (1, 8192, 2048, False)]
N = 12
if sys.argv[1] == "run":
    import ctypes as C, torch
    sys.path.insert(0, "/root/repo")
    from vibevoice_rocm_amd import _lib as L
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    for (m, n, k, dual) in SHAPES:
        x = torch.randn(m, k, device="cuda")
        ws = [((torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16(), (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()) for _ in range(N)]
        out = torch.zeros(m, n, device="cuda")
        torch.cuda.synchronize()
        a = L.LinArgs()
        a.x, a.ldx, a.m, a.n, a.k, a.wdt, a.out, a.ldo = x.data_ptr(), k, m, n, k, L.VV_BF16, out.data_ptr(), n
        if dual: a.act = 2
        for i in range(N):
            a.w = ws[i][0].data_ptr()
            if dual: a.w2 = ws[i][1].data_ptr()
            L.check(lib.vv_linear(C.byref(a), s), "lin")
        torch.cuda.synchronize()
else:
    import csv
    vals = []
    with open(sys.argv[2]) as f:
        rows = [r for r in csv.DictReader(f) if r.get("Counter_Name") == "FETCH_SIZE" and "gemv_stream_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rows) == N * len(SHAPES), (len(rows), N * len(SHAPES))
    out = {}
    for i, (m, n, k, dual) in enumerate(SHAPES):
        seg = rows[i * N: (i + 1) * N]
        kb = sum(float(r["Counter_Value"]) for r in seg) / N
        out[f"{m}x{n}x{k}"] = int(round(2 * 1024 * kb))
        print(f"m={m} n={n} k={k} dual={dual}: {out[f'{m}x{n}x{k}']/1e6:8.2f} MB fetched per launch, weights {n*k*2*(2 if dual else 1)/1e6:8.2f} MB")
    out["_method"] = ("rocprofv3 --pmc FETCH_SIZE (own pass, with --kernel-trace only) over tools/pmc_shapes.py; FETCH_SIZE is in KB and on gfx950 counts 64 B "
                      "per 128-B request for wide coalesced loads, so bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); averaged per launch")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
