"""VibeVoiceProcessor with the reference's call surface (vibevoice/processor/vibevoice_processor.py) on the host:
script parsing (:581-616), prompt assembly (_process_single :231-289, _create_voice_prompt :391-444), left padding
(_batch_encode :291-389), speech padding + masks (prepare_speech_inputs :446-494), AudioNormalizer
(vibevoice_tokenizer_processor.py:19-87) and save_audio (:352-457, 16-bit PCM via the stdlib `wave` module since
soundfile/librosa are not in this image).  CPU-side, off the GPU path.
"""
from __future__ import annotations

import math
import os
import re
import wave
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch


class AudioNormalizer:
    def __init__(self, target_dB_FS: float = -25, eps: float = 1e-6):
        self.target_dB_FS, self.eps = target_dB_FS, eps

    def __call__(self, audio: np.ndarray) -> np.ndarray:
        rms = np.sqrt(np.mean(audio ** 2))
        audio = audio * (10 ** (self.target_dB_FS / 20) / (rms + self.eps))
        max_val = np.max(np.abs(audio))
        return audio / (max_val + self.eps) if max_val > 1.0 else audio / 1.0


def load_wav(path: str, target_sr: int = 24000) -> np.ndarray:
    """Mono float32 in [-1, 1]; linear resampling when the file's rate differs (the reference uses librosa)."""
    with wave.open(path, "rb") as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {sw}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    if sr != target_sr:
        n_out = int(round(len(x) * target_sr / sr))
        x = np.interp(np.linspace(0, len(x) - 1, n_out), np.arange(len(x)), x).astype(np.float32)
    return x


class SyntheticTokenizer:
    """Byte-level stand-in for the Qwen2.5 BPE vocabulary, which cannot be loaded offline.  Text bytes
    map to ids 0..255; the four control ids sit at the top of the model's vocabulary.  A real run passes the
    reference's tokenizer object instead — generate() only reads the *_id attributes."""

    def __init__(self, vocab_size: int):
        self.vocab_size = vocab_size
        self.speech_start_id = vocab_size - 4
        self.speech_end_id = vocab_size - 3
        self.speech_diffusion_id = vocab_size - 2
        self.eos_token_id = self.eos_id = vocab_size - 1
        self.pad_id = self.pad_token_id = vocab_size - 5
        self.bos_token_id = None

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        lim = min(256, self.vocab_size - 8)
        return [b % lim for b in text.encode("utf-8")]


class VibeVoiceProcessor:
    def __init__(self, tokenizer=None, audio_processor=None, speech_tok_compress_ratio: int = 3200, db_normalize: bool = True, **kw):
        self.tokenizer = tokenizer
        self.audio_processor = audio_processor
        self.speech_tok_compress_ratio = speech_tok_compress_ratio
        self.db_normalize = db_normalize
        self.audio_normalizer = AudioNormalizer() if db_normalize else None
        self.system_prompt = " Transform the text provided by various speakers into speech output, utilizing the distinct voice of each respective speaker.\n"

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, tokenizer=None, **kw):
        import json
        cfg_path = os.path.join(str(pretrained_model_name_or_path), "preprocessor_config.json")
        cfg = {}
        if os.path.exists(cfg_path):
            with open(cfg_path) as f:
                cfg = json.load(f)
        if tokenizer is None:
            tok_dir = str(pretrained_model_name_or_path)
            if os.path.exists(os.path.join(tok_dir, "tokenizer.json")) or os.path.exists(os.path.join(tok_dir, "vocab.json")):
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(tok_dir)
                tokenizer.speech_start_id = tokenizer.convert_tokens_to_ids("<|vision_start|>")
                tokenizer.speech_end_id = tokenizer.convert_tokens_to_ids("<|vision_end|>")
                tokenizer.speech_diffusion_id = tokenizer.convert_tokens_to_ids("<|vision_pad|>")
                tokenizer.pad_id = tokenizer.convert_tokens_to_ids("<|image_pad|>")
            else:
                raise OSError("no tokenizer files next to the checkpoint and no hub access: pass tokenizer=")
        return cls(tokenizer=tokenizer, speech_tok_compress_ratio=cfg.get("speech_tok_compress_ratio", 3200),
                   db_normalize=cfg.get("db_normalize", True))

    # ---- script ------------------------------------------------------------------------------------------------
    @staticmethod
    def _parse_script(script: str) -> List[Tuple[int, str]]:
        parsed, ids = [], []
        for line in script.strip().split("\n"):
            if not line.strip():
                continue
            m = re.match(r"^Speaker\s+(\d+)\s*:\s*(.*)$", line.strip(), re.IGNORECASE)
            if m:
                parsed.append((int(m.group(1)), " " + m.group(2).strip()))
                ids.append(int(m.group(1)))
        if not parsed:
            raise ValueError("No valid speaker lines found in script")
        if min(ids) > 0:
            parsed = [(i - 1, t) for i, t in parsed]
        return parsed

    @staticmethod
    def _convert_json_to_script(json_file: str) -> str:
        """[{"speaker": "1", "text": "..."}, ...] -> "Speaker 1: ..." lines; malformed entries are skipped (:496-541)."""
        import json
        with open(json_file, "r", encoding="utf-8") as f:
            data = json.load(f)
        if not isinstance(data, list):
            raise ValueError("JSON file must contain a list of speaker entries")
        lines = []
        for item in data:
            if not isinstance(item, dict) or item.get("speaker") is None or item.get("text") is None:
                continue
            try:
                speaker_id = int(item["speaker"])
            except (ValueError, TypeError):
                continue
            text = item["text"].strip()
            if text:
                lines.append(f"Speaker {speaker_id}: {text}")
        if not lines:
            raise ValueError("No valid entries found in JSON file")
        return "\n".join(lines)

    @staticmethod
    def _convert_text_to_script(text_file: str) -> str:
        """"Speaker X: text" lines are kept, plain lines go to Speaker 1, empty lines / empty texts are dropped (:543-580)."""
        with open(text_file, "r", encoding="utf-8") as f:
            raw = f.readlines()
        lines = []
        for line in raw:
            line = line.strip()
            if not line:
                continue
            m = re.match(r"^Speaker\s+(\d+)\s*:\s*(.*)$", line, re.IGNORECASE)
            if m:
                if m.group(2).strip():
                    lines.append(f"Speaker {int(m.group(1))}: {m.group(2).strip()}")
            else:
                lines.append(f"Speaker 1: {line}")
        if not lines:
            raise ValueError("No valid content found in text file")
        return "\n".join(lines)

    def _create_voice_prompt(self, speaker_samples):
        tok = self.tokenizer
        tokens = tok.encode(" Voice input:\n", add_special_tokens=False)
        speech_inputs, masks = [], [False] * len(tokens)
        for speaker_id, audio in enumerate(speaker_samples):
            prefix = tok.encode(f" Speaker {speaker_id}:", add_special_tokens=False)
            wav = load_wav(audio) if isinstance(audio, str) else np.array(audio, dtype=np.float32)
            if self.db_normalize and self.audio_normalizer:
                wav = self.audio_normalizer(wav)
            n = math.ceil(wav.shape[0] / self.speech_tok_compress_ratio)
            nl = tok.encode("\n", add_special_tokens=False)
            tokens += prefix + [tok.speech_start_id] + [tok.speech_diffusion_id] * n + [tok.speech_end_id] + nl
            masks += [False] * len(prefix) + [False] + [True] * n + [False] + [False] * len(nl)
            speech_inputs.append(wav.astype(np.float32))
        return tokens, speech_inputs, masks

    def _process_single(self, text: str, voice_samples=None) -> Dict[str, Any]:
        if not isinstance(text, str):
            raise ValueError(f"Could not process input text: {text}")
        if text.endswith(".json") and os.path.exists(text):                  # :242-244
            text = self._convert_json_to_script(text)
        elif text.endswith(".txt") and os.path.exists(text):
            text = self._convert_text_to_script(text)
        parsed = self._parse_script(text)
        speakers = list(set(s for s, _ in parsed))
        tok = self.tokenizer
        full = tok.encode(self.system_prompt)
        mask = [False] * len(full)
        speech_inputs = []
        if voice_samples:
            vt, speech_inputs, vm = self._create_voice_prompt(voice_samples[: len(speakers)])
            full += vt
            mask += vm
        t = tok.encode(" Text input:\n", add_special_tokens=False)
        full += t
        mask += [False] * len(t)
        for sid, stext in parsed:
            t = tok.encode(f" Speaker {sid}:{stext}\n", add_special_tokens=False)
            full += t
            mask += [False] * len(t)
        t = tok.encode(" Speech output:\n", add_special_tokens=False) + [tok.speech_start_id]
        full += t
        mask += [False] * len(t)
        return dict(input_ids=full, speech_inputs=speech_inputs or None, speech_input_mask=mask, parsed_script=parsed, all_speakers=speakers)

    def prepare_speech_inputs(self, speech_inputs: List[np.ndarray], return_tensors=None):
        if not speech_inputs:
            return dict(padded_speeches=None, speech_masks=None)
        lens = [math.ceil(s.shape[0] / self.speech_tok_compress_ratio) for s in speech_inputs]
        T = max(s.shape[0] for s in speech_inputs)
        padded = np.zeros((len(speech_inputs), T), dtype=np.float32)
        masks = np.zeros((len(speech_inputs), max(lens)), dtype=np.bool_)
        for i, (s, n) in enumerate(zip(speech_inputs, lens)):
            padded[i, : len(s)] = s
            masks[i, :n] = True
        if return_tensors == "pt":
            return dict(padded_speeches=torch.tensor(padded), speech_masks=torch.tensor(masks))
        return dict(padded_speeches=padded, speech_masks=masks)

    def __call__(self, text=None, voice_samples=None, padding=True, truncation=False, max_length=None, return_tensors=None,
                 return_attention_mask=True, **kw):
        if isinstance(text, str):
            texts, batched = [text], False
        else:
            texts, batched = list(text), True
        if voice_samples is not None:
            vs = [voice_samples] if (not batched or isinstance(voice_samples[0], (str, np.ndarray))) else voice_samples
        else:
            vs = [None] * len(texts)
        encs = [self._process_single(t, v) for t, v in zip(texts, vs)]
        mx = max(len(e["input_ids"]) for e in encs)
        ids, am, sm, speech = [], [], [], []
        for e in encs:
            pad = (mx - len(e["input_ids"])) if padding else 0
            ids.append([self.tokenizer.pad_id] * pad + e["input_ids"])          # LEFT padding (:336)
            am.append([0] * pad + [1] * len(e["input_ids"]))
            sm.append([False] * pad + e["speech_input_mask"])
            if e["speech_inputs"] is not None:
                speech += e["speech_inputs"]
        out: Dict[str, Any] = {}
        if return_tensors is not None:
            out["input_ids"] = torch.tensor(ids, dtype=torch.long)
            if return_attention_mask:
                out["attention_mask"] = torch.tensor(am, dtype=torch.long)
            out["speech_input_mask"] = torch.tensor(sm, dtype=torch.bool)
        else:
            out["input_ids"], out["speech_input_mask"] = ids, sm
            if return_attention_mask:
                out["attention_mask"] = am
        sp = self.prepare_speech_inputs(speech, return_tensors=return_tensors)
        out["speech_tensors"], out["speech_masks"] = sp["padded_speeches"], sp["speech_masks"]
        out["parsed_scripts"] = [e["parsed_script"] for e in encs]
        out["all_speakers_list"] = [e["all_speakers"] for e in encs]
        return out

    # ---- output ------------------------------------------------------------------------------------------------
    def save_audio(self, audio, output_path: str = "output.wav", sampling_rate: int = 24000, normalize: bool = False, **kw):
        if isinstance(audio, torch.Tensor):
            audio = audio.detach().float().cpu().numpy()
        x = np.asarray(audio, dtype=np.float32).reshape(-1)
        if normalize and self.audio_normalizer:
            x = self.audio_normalizer(x)
        pcm = (np.clip(x, -1.0, 1.0) * 32767.0).astype("<i2")
        os.makedirs(os.path.dirname(os.path.abspath(output_path)) or ".", exist_ok=True)
        with wave.open(output_path, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(sampling_rate)
            w.writeframes(pcm.tobytes())
        return [output_path]
