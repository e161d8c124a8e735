"""One long-context LLM decode step configuration for rocprofv3: python tools/mb_attn_one.py <model> <S> <s_max> [attn_gqa]"""
import sys, ctypes as C, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vibevoice_rocm_amd import _lib as L
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
model, S, smax = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = VVConfig.preset(model)
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
eng = m.engine; lib = eng.lib; V = cfg.vocab
if len(sys.argv) > 4:
    lib.vv_tune(b"attn_gqa", int(sys.argv[4]))
eng.begin_sequence(smax, [V-4, V-3, V-2, V-1])
with torch.cuda.stream(eng.stream):
    eng.lens.copy_(torch.tensor([S, S // 3], dtype=torch.int32))
    lens0 = eng.lens.clone()
    for _ in range(10):
        eng._seq_A(V-4, V-2)
        eng.lens.copy_(lens0)
eng.stream.synchronize()
