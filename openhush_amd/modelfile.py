"""ggml `ggml-*.bin` Whisper model files: writer (synthetic models) and reader (checks).

Format as described in SURVEY.md Appendix A (the only format the reference downloads:
reference src/engine/whisper.rs:71-102).  Layout:
  u32 magic 0x67676d6c
  11 x i32 hparams  (n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer,
                     n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype)
  i32 n_mel, i32 n_fft(=201), n_mel*n_fft f32 mel filters
  i32 n_tokens, then n_tokens x { u32 len, bytes }
  tensors until EOF: { i32 n_dims, i32 name_len, i32 ttype (0 f32, 1 f16),
                       i32 dims[n_dims] (fastest-varying first), name bytes, raw data }
Parity of this reader/writer with real files is unpinned (no real file exists offline).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Tuple

import numpy as np

from . import synth

GGML_MAGIC = 0x67676D6C


def synthetic_vocab(hp: synth.HParams) -> List[bytes]:
    """A stand-in vocabulary: printable byte strings for text tokens.  The stock files carry
    only the text tokens (< eot) plus nothing for specials; whisper.cpp synthesises names for the
    rest.  We write n_vocab - (specials) entries exactly like the stock converter (eot index)."""
    n_text = 50257 if hp.is_multilingual else 50256
    words = []
    for i in range(n_text):
        words.append((" w%d" % i).encode("ascii"))
    words[220] = b" "   # the real multilingual vocabulary has " " at 220 (suppress_blank target)
    return words


def write_synthetic_model(path: str, hp: synth.HParams, seed: int = 1234) -> None:
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(struct.pack("<I", GGML_MAGIC))
        f.write(struct.pack("<11i", *hp.as_list()))
        filt = synth.mel_filterbank(hp.n_mels)
        f.write(struct.pack("<2i", hp.n_mels, synth.N_FREQ))
        f.write(filt.astype("<f4").tobytes())
        vocab = synthetic_vocab(hp)
        f.write(struct.pack("<i", len(vocab)))
        for w in vocab:
            f.write(struct.pack("<I", len(w)))
            f.write(w)
        # the generator is element-wise numpy (the GIL is released inside it): tensors are made by a few threads, a
        # bounded number ahead of the writer, and written in the file's order (large-v3 = 3.1 GB: 90 s -> 15 s)
        import concurrent.futures as cf
        specs = synth.tensor_specs(hp)

        def make(spec):
            as_f16 = spec.f16 and hp.ftype == 1
            return synth.gen_tensor(seed, spec, hp).astype("<f2" if as_f16 else "<f4"), as_f16

        workers = max(1, min(8, len(os.sched_getaffinity(0))))
        with cf.ThreadPoolExecutor(max_workers=workers) as ex:
            pending = []
            it = iter(specs)
            for spec in it:
                pending.append((spec, ex.submit(make, spec)))
                if len(pending) < 2 * workers:
                    continue
                _write_tensor(f, *pending.pop(0))
            while pending:
                _write_tensor(f, *pending.pop(0))
    os.replace(tmp, path)


def _write_tensor(f, spec, fut) -> None:
    arr, as_f16 = fut.result()
    name = spec.name.encode("ascii")
    dims = list(reversed(spec.shape))
    f.write(struct.pack("<3i", len(dims), len(name), 1 if as_f16 else 0))
    f.write(struct.pack("<%di" % len(dims), *dims))
    f.write(name)
    f.write(arr.tobytes() if arr.size < (1 << 20) else memoryview(arr).cast("B"))


def read_model(path: str) -> Tuple[synth.HParams, np.ndarray, List[bytes], Dict[str, np.ndarray]]:
    with open(path, "rb") as f:
        buf = f.read()
    off = 0
    (magic,) = struct.unpack_from("<I", buf, off); off += 4
    if magic != GGML_MAGIC:
        raise ValueError("bad magic 0x%08x" % magic)
    vals = struct.unpack_from("<11i", buf, off); off += 44
    hp = synth.HParams(*vals)
    n_mel, n_fft = struct.unpack_from("<2i", buf, off); off += 8
    filt = np.frombuffer(buf, "<f4", n_mel * n_fft, off).reshape(n_mel, n_fft).copy(); off += 4 * n_mel * n_fft
    (n_tok,) = struct.unpack_from("<i", buf, off); off += 4
    vocab = []
    for _ in range(n_tok):
        (ln,) = struct.unpack_from("<I", buf, off); off += 4
        vocab.append(bytes(buf[off:off + ln])); off += ln
    tensors: Dict[str, np.ndarray] = {}
    while off < len(buf):
        n_dims, name_len, ttype = struct.unpack_from("<3i", buf, off); off += 12
        dims = struct.unpack_from("<%di" % n_dims, buf, off); off += 4 * n_dims
        name = bytes(buf[off:off + name_len]).decode("ascii"); off += name_len
        shape = tuple(reversed(dims))
        n = int(np.prod(shape))
        dt = "<f2" if ttype == 1 else "<f4"
        arr = np.frombuffer(buf, dt, n, off).reshape(shape); off += n * (2 if ttype == 1 else 4)
        tensors[name] = arr.astype(np.float32)
    return hp, filt, vocab, tensors
