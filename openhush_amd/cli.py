"""`openhush transcribe FILE` plumbing over the MI355X engine (BASELINE.json config #1).

Mirrors the reference's one-shot CLI arm (reference src/main.rs:974-1077) and its WAV input contract
(reference src/input/audio.rs:348-434, `load_wav_file`): integer samples / 2^(bits-1), channel average, pad
with silence to 1.1 s; prints the same JSON fields as `--format json` (src/main.rs:1054-1066).
Only 16 kHz input is accepted here: the reference resamples with the `rubato` sinc resampler, which belongs to
the DSP front end (out of scope).

    python -m openhush_amd.cli transcribe audio.wav --model-path /path/ggml-small.bin [--format json]

Long recordings over several GPUs of one node (BASELINE.json config #4): launch the same command under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m openhush_amd.cli transcribe ...`;
the 30 s windows are dealt round-robin to the ranks (one GPU each), the token ids are gathered on rank 0 (RCCL), which
prints the result (openhush_amd/shard.py).
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import sys
import time
import wave

import numpy as np

SAMPLE_RATE = 16000
WHISPER_MIN_DURATION_SECS = 1.1   # reference src/input/audio.rs:34


def load_wav_file(path: str) -> np.ndarray:
    """float32 mono 16 kHz samples, padded to 1.1 s like the reference's load_wav_file."""
    with wave.open(path, "rb") as w:
        rate, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if rate != SAMPLE_RATE:
        raise ValueError(f"{path}: {rate} Hz input needs the resampling front end (only {SAMPLE_RATE} Hz is accepted here)")
    if width == 1:      # 8-bit WAV is unsigned; hound yields i8 = u8 - 128
        s = (np.frombuffer(raw, np.uint8).astype(np.int32) - 128).astype(np.float32) / np.float32(1 << 7)
    elif width == 2:
        s = np.frombuffer(raw, "<i2").astype(np.float32) / np.float32(1 << 15)
    elif width == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        s = v.astype(np.float32) / np.float32(1 << 23)
    elif width == 4:
        s = np.frombuffer(raw, "<i4").astype(np.float32) / np.float32(2.0 ** 31)
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    if ch > 1:   # average the channels (reference :384-391)
        s = (s[: len(s) // ch * ch].reshape(-1, ch).sum(axis=1, dtype=np.float32) / np.float32(ch)).astype(np.float32)
    need = int(np.float32(SAMPLE_RATE) * np.float32(WHISPER_MIN_DURATION_SECS))
    if len(s) / SAMPLE_RATE < WHISPER_MIN_DURATION_SECS:
        s = np.concatenate([s, np.zeros(need - len(s), np.float32)])
    return np.ascontiguousarray(s, dtype=np.float32)


def _transcribe_ranks(args, audio) -> int:
    """one process per GPU under torch.distributed.run: shard the windows, gather the tokens, rank 0 prints"""
    import torch
    import torch.distributed as dist
    from . import engine as E, shard
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    E.validate_audio(audio.samples, audio.sample_rate)
    # rank 0 reads the file, the packed weights reach the other GPUs in one RCCL broadcast
    ctx = shard.load_model_broadcast(args.model_path, dist, world, rank, local, {"auto": E.OHW_DTYPE_AUTO, "bf16": E.OHW_DTYPE_BF16, "f16": E.OHW_DTYPE_F16}[args.dtype])
    p = ctx.default_params()
    if args.language != "auto":
        p.lang_id = E.lang_code_to_id(args.language)
    p.translate = 1 if args.translate else 0
    n_win = (len(audio.samples) + E.CHUNK_SAMPLES - 1) // E.CHUNK_SAMPLES
    t1 = time.perf_counter()
    by_index = args.mel == "recording"
    runner = shard.recording_window_runner(ctx, args.max_batch, audio.samples, p) if by_index else shard.engine_window_runner(ctx, args.max_batch, p)
    wins = shard.transcribe_sharded(runner, audio.samples, n_win, ctx.hp.n_text_ctx, dist, world, rank, torch.device("cuda", local),
                                    by_index=by_index)
    dt = time.perf_counter() - t1
    if rank == 0:
        text = b"".join(ctx.token_text(t) for w in wins for t in w if t < ctx.tok.eot).decode("utf-8", "replace").strip()
        lang = "en" if args.language == "auto" else args.language
        name = args.model or args.model_path.rsplit("/", 1)[-1].replace("ggml-", "").rsplit(".", 1)[0]
        if args.format == "json":
            print(json.dumps({"text": text, "language": lang, "duration_ms": int(dt * 1e3), "audio_duration_secs": audio.duration_secs(),
                              "transcription_time_ms": int(dt * 1e3), "real_time_factor": dt / audio.duration_secs(),
                              "model": name.lower(), "gpus": world}, indent=2))
        else:
            print("\n--- Transcription ---")
            print(text)
            print("---")
            print(f"\nTime: {dt * 1e3:.0f}ms (RTF: {dt / audio.duration_secs():.3f}x) on {world} GPUs")
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="openhush_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    t = sub.add_parser("transcribe")
    t.add_argument("file")
    t.add_argument("--model-path", required=True, help="ggml-*.bin model file")
    t.add_argument("--model", default=None, help="name printed in the JSON (default: derived from the file name)")
    t.add_argument("--language", default="auto")
    t.add_argument("--translate", action="store_true")
    t.add_argument("--format", default="text", choices=["text", "json"])
    t.add_argument("--device", default="hip:0")
    t.add_argument("--dtype", default="auto", choices=["auto", "bf16", "f16"],
                   help="auto (default): the model file's own precision - f16 for the stock ggml files (ftype 1), whose weights then stay exact")
    t.add_argument("--max-batch", type=int, default=8)
    t.add_argument("--mel", default="window", choices=["window", "recording"],
                   help="window (default): every 30 s cut is its own call; recording: the cuts are taken from the spectrogram of the whole "
                        "recording (one clamp maximum, real samples across the 30 s marks), as whisper.cpp computes it for one call")
    args = ap.parse_args(argv)

    from . import engine as E
    if not os.path.isfile(args.file):            # reference src/main.rs:985-987 and tests/cli_integration.rs:262-269
        print(f"error: File not found: {args.file}", file=sys.stderr)
        return 1
    try:
        audio = E.AudioBuffer(load_wav_file(args.file), SAMPLE_RATE)
    except (ValueError, wave.Error, EOFError) as ex:
        print(f"error: {ex}", file=sys.stderr)
        return 1
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return _transcribe_ranks(args, audio)
    use_gpu = args.device.lower() != "cpu"                    # reference src/main.rs:1037
    dev = int(args.device.split(":")[1]) if ":" in args.device else 0
    t0 = time.perf_counter()
    eng = E.WhisperEngine.new(args.model_path, args.language, args.translate, use_gpu, dev,
                              {"auto": E.OHW_DTYPE_AUTO, "bf16": E.OHW_DTYPE_BF16, "f16": E.OHW_DTYPE_F16}[args.dtype], args.max_batch)
    print(f"Model loaded in {1e3 * (time.perf_counter() - t0):.0f}ms", file=sys.stderr)
    if args.mel == "recording":
        eng.set_window_mode(E.OHW_WINDOW_FIXED_RECORDING_MEL)
    t1 = time.perf_counter()
    res = eng.transcribe(audio)
    dt = time.perf_counter() - t1
    rtf = dt / audio.duration_secs()
    name = args.model or args.model_path.rsplit("/", 1)[-1].replace("ggml-", "").rsplit(".", 1)[0]
    if args.format == "json":
        print(json.dumps({"text": res.text, "language": res.language, "duration_ms": res.duration_ms,
                          "audio_duration_secs": audio.duration_secs(), "transcription_time_ms": int(dt * 1e3),
                          "real_time_factor": rtf, "model": name.lower()}, indent=2))
    else:
        print("\n--- Transcription ---")
        print(res.text)
        print("---")
        print(f"\nTime: {dt * 1e3:.0f}ms (RTF: {rtf:.3f}x)")
    eng.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
