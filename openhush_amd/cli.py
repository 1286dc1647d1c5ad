"""`openhush transcribe FILE` plumbing over the MI355X engine (BASELINE.json config #1).

Mirrors the reference's one-shot CLI arm (reference src/main.rs:974-1077) and its WAV input contract
(reference src/input/audio.rs:348-434, `load_wav_file`): integer samples / 2^(bits-1), channel average, pad
with silence to 1.1 s; prints the same JSON fields as `--format json` (src/main.rs:1054-1066).
Any sample rate is accepted and resampled to 16 kHz (`--resampling-quality high` = the sinc resampler the reference's
default picks, `low` = linear: reference :394-407, config.audio.resampling_quality).

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

import numpy as np

SAMPLE_RATE = 16000
WHISPER_MIN_DURATION_SECS = 1.1   # reference src/input/audio.rs:34


def _read_riff(path: str):
    """(rate, channels, float32 interleaved samples) of a RIFF/WAVE file: PCM integers of 8 / 16 / 24 / 32 bits scaled by
    2^(bits-1) and 32-bit IEEE floats as they are - the two hound::SampleFormat arms of the reference (:369-382)"""
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: Failed to open WAV file: no RIFF/WAVE header")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        cid, size = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
        body = blob[pos + 8:pos + 8 + size]
        if cid == b"fmt " and len(body) >= 16:
            fmt = body
        elif cid == b"data":
            data = body                      # a truncated file keeps what is there (hound's filter_map(Result::ok))
            if fmt is not None:
                break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: Failed to open WAV file: fmt or data chunk missing")
    tag, ch, rate, _, _, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == 0xFFFE and len(fmt) >= 26:     # WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes are the real tag
        tag = struct.unpack_from("<H", fmt, 24)[0]
    if ch < 1 or rate < 1:
        raise ValueError(f"{path}: Failed to open WAV file: bad channel count or rate")
    width = bits // 8
    data = data[:len(data) // width * width] if width else data
    if tag == 3 and bits == 32:
        s = np.frombuffer(data, "<f4").astype(np.float32)
    elif tag == 1 and width == 1:      # 8-bit WAV is unsigned; hound yields i8 = u8 - 128
        s = (np.frombuffer(data, np.uint8).astype(np.int32) - 128).astype(np.float32) / np.float32(1 << 7)
    elif tag == 1 and width == 2:
        s = np.frombuffer(data, "<i2").astype(np.float32) / np.float32(1 << 15)
    elif tag == 1 and width == 3:
        b = np.frombuffer(data, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        s = v.astype(np.float32) / np.float32(1 << 23)
    elif tag == 1 and width == 4:
        s = np.frombuffer(data, "<i4").astype(np.float32) / np.float32(2.0 ** 31)
    else:
        raise ValueError(f"{path}: unsupported WAV format (tag {tag}, {bits} bits)")
    return rate, ch, s


def load_wav_file(path: str, quality: str = "high") -> np.ndarray:
    """float32 mono 16 kHz samples like the reference's load_wav_file (src/input/audio.rs:348-434): any rate (resampled to
    16 kHz: `high` = the sinc resampler of :1007-1095, `low` = linear :972-990, as config.audio.resampling_quality picks -
    the reference's default is high), any of its sample formats, channels averaged, padded with silence to 1.1 s."""
    rate, ch, s = _read_riff(path)
    if ch > 1:   # average the channels (reference :384-391)
        s = (s[: len(s) // ch * ch].reshape(-1, ch).sum(axis=1, dtype=np.float32) / np.float32(ch)).astype(np.float32)
    if rate != SAMPLE_RATE:              # reference :394-407
        from . import engine as E
        s = E.resample_sinc(s, rate, SAMPLE_RATE) if quality == "high" else E.resample_linear(s, rate, SAMPLE_RATE)
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
    t.add_argument("--resampling-quality", default="high", choices=["low", "high"],
                   help="for files that are not 16 kHz (the reference's config.audio.resampling_quality, default high = sinc)")
    t.add_argument("--mel", default="window", choices=["window", "recording"],
                   help="window (default): every 30 s cut is its own call; recording: the cuts are taken from the spectrogram of the whole "
                        "recording (one clamp maximum, real samples across the 30 s marks), as whisper.cpp computes it for one call")
    args = ap.parse_args(argv)

    from . import engine as E
    if not os.path.isfile(args.file):            # reference src/main.rs:985-987 and tests/cli_integration.rs:262-269
        print(f"error: File not found: {args.file}", file=sys.stderr)
        return 1
    try:
        audio = E.AudioBuffer(load_wav_file(args.file, args.resampling_quality), SAMPLE_RATE)
    except (ValueError, struct.error, EOFError) as ex:
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
