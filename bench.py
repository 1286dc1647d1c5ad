#!/usr/bin/env python3
"""bench.py — throughput of the Whisper hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole path over one batch of synthetic 30 s windows per GPU:
log-mel -> encoder -> cross K/V -> greedy decode (EOT suppressed, exactly --tokens tokens per window,
SURVEY.md 8d) with the PCM already resident in HBM.  Workload at every N: large-v3 dimensions,
--batch windows per GPU (weak scaling: independent windows, no data-path collective; the only
communication is the gather of token ids to rank 0 after each step).

    python bench.py                       # 1 GPU, defaults finish in a few minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP events on the launch stream) and `cpu_baseline` (oracle timed on this host's cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_HBM_GBS = 8000.0
PEAK_MFMA_TFLOPS = 2500.0   # dense bf16 / f16

PROF_NAMES = {1: "gemm256_kernel (encoder / cross-KV MFMA GEMM)", 2: "encoder_attention_kernel", 3: "cross_attn_kernel (decoder)",
              4: "dec_gemm_kernel<RESID> (decoder out-projections, mlp.2)", 5: "dec_gemm_kernel<QKV,LN>",
              6: "dec_gemm_kernel<BIAS,LN> (cross query)", 7: "dec_gemm_kernel<GELU,LN> (mlp.0)", 8: "dec_gemm_kernel<LOGITS>"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=100)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--profile-class", type=int, default=0, help="0 = pick the class with the largest total time")
    ap.add_argument("--streams", type=int, default=1, help="split the batch over this many concurrent states/streams")
    ap.add_argument("--pipeline", type=int, default=96,
                    help="with --phases 0 or 1: N > 0 puts two batches in flight - mel/encoder/cross-K/V of batch i+1 on N compute units "
                         "beside the decoder of batch i on the rest (round 1's schedule); 0: one batch after the other")
    ap.add_argument("--phases", type=int, default=4,
                    help="G > 1 (default 4, the engine's LANES schedule): the front ends of G lanes (--merge batches each) run one after the other on "
                         "every CU, then G decodes side by side on G disjoint CU sets (one's latency-bound GEMM chain under the others' K/V "
                         "streams); 0 or 1: see --pipeline")
    ap.add_argument("--merge", type=int, default=3,
                    help="with --phases: batches (steps) one lane takes through ONE front-end pass and ONE decode of merge x batch "
                         "windows - the decoder streams its weights once per token whatever its batch, and 96 windows fill the "
                         "encoder GEMMs' last round of 256-row tiles")
    ap.add_argument("--pool", action="store_true",
                    help="ONE process drives --gpus devices through the C ABI's ohw_pool_* (ncclCommInitAll + one broadcast of the weight "
                         "arena, windows dealt round-robin, host gather): the design BASELINE.json's north_star describes.  Launch plainly "
                         "(python bench.py --pool --gpus N), not under torch.distributed.run; host PCM in (PCIe-inclusive)")
    ap.add_argument("--pool-devices", default="", help="with --pool: explicit device list, e.g. 0,0 to rehearse two engines on one card")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch latency figures (large-v3 B = 1, config #2, config #5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=0, help="decode tokens in the CPU sample (0 = same as --tokens)")
    ap.add_argument("--cpu-windows", type=int, default=2, help="30 s windows in the CPU sample")
    args = ap.parse_args()

    import torch
    from openhush_amd import engine as E, shard, synth

    if args.pool:
        return pool_bench(args, torch, E, synth)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a ONE-GPU box: OHW_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo (RCCL needs
    # one device per rank).  Never set by the driver; numbers from such a run mean nothing.
    rehearse = os.environ.get("OHW_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or "RANK" in os.environ    # under torch.distributed.run the collectives run even at N = 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # keep stdout to the ONE JSON line: RCCL prints a version banner there when NCCL_DEBUG is VERSION/INFO
        if "OHW_NCCL_DEBUG" in os.environ:
            os.environ["NCCL_DEBUG"] = os.environ["OHW_NCCL_DEBUG"]
        else:
            os.environ.pop("NCCL_DEBUG", None)
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    hp = synth.PRESETS[args.model]
    dtype = E.OHW_DTYPE_BF16 if args.dtype == "bf16" else E.OHW_DTYPE_F16
    B = args.batch
    if use_dist:
        # the multi-GPU load path: rank 0 makes the model (here: procedural weights; in production: reads the ggml file),
        # its resident weight blob (3.1 GB) goes to the other ranks in ONE RCCL broadcast over xGMI, they import it into a
        # shell context (include/ohw.h, ohw_ctx_blob_*; shard.load_model_broadcast does the same from a file).  Not timed.
        ctx = E.Context.synthetic(hp.as_list(), 1234, local_rank, dtype) if rank == 0 else E.Context.shell(hp.as_list(), local_rank, dtype)
        nblob = ctx.blob_size()
        blob = torch.empty(nblob, dtype=torch.uint8, device=torch.device("cuda", local_rank))
        if rank == 0:
            ctx.export_blob(blob.data_ptr(), nblob)
        dist.broadcast(blob, src=0)
        torch.cuda.synchronize()
        if rank != 0:
            ctx.import_blob(blob.data_ptr(), nblob)
        del blob
    else:
        ctx = E.Context.synthetic(hp.as_list(), 1234, local_rank, dtype)
    # the batch may be split over several concurrent states (each with its own HIP stream): the
    # latency-bound decoder kernels of one sub-batch then overlap the HBM-bound ones of another
    S = max(1, args.streams)
    if B % S != 0:
        raise SystemExit("--batch must be divisible by --streams")
    Bs = B // S
    states = [E.State(ctx, Bs) for _ in range(S)]
    st = states[0]

    # synthetic 16 kHz audio: window ids are global so that every rank transcribes different audio
    pcm_host = np.stack([synth.synth_audio(rank * B + b) for b in range(B)])
    pcm = torch.from_numpy(pcm_host).cuda()
    n_samples = [synth.CHUNK_SAMPLES] * Bs
    p = ctx.default_params()
    p.force_len = args.tokens

    dev = torch.device("cuda", local_rank)
    import threading

    def run_state(i, out):
        s_ = states[i]
        s_.mel_device(pcm.data_ptr() + i * Bs * pcm.shape[1] * 4, pcm.shape[1], n_samples, E.OHW_MEL_ZERO_TAIL)
        s_.encode(Bs)
        out[i], _ = s_.greedy(Bs, p)

    def step():
        out = [None] * S
        if S == 1:
            run_state(0, out)
        else:   # ctypes releases the GIL inside the library: the S host threads enqueue concurrently
            th = [threading.Thread(target=run_state, args=(i, out)) for i in range(S)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        toks = [t for o in out for t in o]
        if use_dist:    # the path's only communication: token ids of every rank's windows to rank 0 (KBs, RCCL)
            shard.gather_tokens(shard.pack_tokens(toks, args.tokens), dist, world, rank, dev, force=True)
        return toks

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(n):
        toks = None
        for _ in range(n):
            toks = step()
        return toks

    pipe = None
    G = max(0, args.phases)
    MG = max(1, args.merge)
    if rehearse:
        MG = 1        # the ranks of a rehearsal share ONE GPU's memory: lanes of one batch (a 96-window lane holds 37 GB)
    if G > 1 and S == 1:
        n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
        try:
            full = E.Stream(local_rank, 0, 0)
            dss = [E.Stream(local_rank, l * (n_cu // G), n_cu // G) for l in range(G)]
            es = ds = full
            pst = [E.State(ctx, B * MG) for _ in range(G)]
            # every window in flight is its own synthetic recording: lane j's decode batch = windows [j * MG * B, (j + 1) * MG * B)
            # of this rank's G * MG * B (window ids are global: rank * G * MG * B + ...)
            n_fl = G * MG * B
            pcm_all = torch.from_numpy(np.stack([synth.synth_audio(rank * n_fl + w) for w in range(n_fl)])).cuda() if n_fl > B else pcm
            pipe = {"schedule": "lanes", "decoders_side_by_side": G, "batches_per_decode": MG, "decoder_cus": n_cu // G,
                    "batches_in_flight": G * MG}
        except E.WhisperError as ex:
            print(f"[bench] CU-masked streams unavailable ({ex}); running one batch after the other", file=sys.stderr, flush=True)
            pipe = None
    if pipe and G > 1:
        def run_steps(n):           # noqa: F811
            out = [None] * n
            err = []
            for g0 in range(0, n, G * MG):
                grp = list(range(g0, min(n, g0 + G * MG)))
                # lane j decodes steps grp[j * MG : (j + 1) * MG] as ONE batch; their front ends (B windows each) run one after the
                # other on every CU and write their cross K/V into that lane's decode batch (ohw_encode_slice)
                # (a short last group is dealt evenly: two half-size decodes side by side beat one full-size decode alone)
                cnt = [len(grp) // G + (1 if j < len(grp) % G else 0) for j in range(G)]
                lanes, at = [], 0
                for c in cnt:
                    if c:
                        lanes.append(grp[at:at + c])
                    at += c
                for j, steps_j in enumerate(lanes):
                    s_ = pst[j]
                    s_.set_stream(full.ptr)
                    # ONE front-end pass over the lane's len(steps_j) x B windows (the engine does the same): 96 windows fill
                    # the last round of the encoder GEMMs' 256-row tiles where 32 leave it two thirds empty
                    c = len(steps_j)
                    s_.mel_device(pcm_all.data_ptr() + j * MG * B * pcm_all.shape[1] * 4, pcm_all.shape[1], n_samples * c, E.OHW_MEL_ZERO_TAIL)
                    s_.encode(c * B)
                # the host waits for the group's front ends before it starts the lane threads: lanes that begin to enqueue
                # their decode while the front ends still run cost 6 % of a step (283.9 against 261.7 - 268.2 ms, measured)
                dbg = os.environ.get("OHW_BENCH_DEBUG") == "1"
                tg0 = time.perf_counter(); full.sync(); tg1 = time.perf_counter()

                lane_ms = [0.0] * len(lanes)

                def lane(j, steps_j):
                    try:
                        tl0 = time.perf_counter()
                        toks_j, _ = pst[j].greedy(len(steps_j) * B, p)
                        lane_ms[j] = 1e3 * (time.perf_counter() - tl0)
                        for k, i in enumerate(steps_j):
                            out[i] = toks_j[k * B:(k + 1) * B]
                    except Exception as ex:      # noqa: BLE001
                        err.append(ex)
                if len(lanes) == 1:
                    lane(0, lanes[0])
                else:
                    for j in range(len(lanes)):
                        dss[j].wait(full)
                        pst[j].set_stream(dss[j].ptr)
                    th = [threading.Thread(target=lane, args=(j, sj)) for j, sj in enumerate(lanes)]
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    for j in range(len(lanes)):
                        full.wait(dss[j])            # the next group's front ends start after every decode of this one
                if dbg:
                    print(f"[bench] group {grp}: front ends drained after {1e3 * (tg1 - tg0):.1f} ms of waiting, decodes {1e3 * (time.perf_counter() - tg1):.1f} ms "
                          f"(lanes: {', '.join(f'{x:.0f}' for x in lane_ms)})", file=sys.stderr, flush=True)
                if err:
                    raise err[0]
            if use_dist:
                for toks_i in out:
                    shard.gather_tokens(shard.pack_tokens(toks_i, args.tokens), dist, world, rank, dev, force=True)
            return out[-1] if n > 0 else None
    elif args.pipeline > 0 and S == 1:
        # Batches in flight on disjoint CU sets (include/ohw.h, ohw_stream_create): while batch i decodes (HBM- and
        # latency-bound, MFMA idle) batch i+1 runs mel + encoder + cross-K/V (MFMA-bound) - round 1's schedule.  A step is still one whole
        # pass over one batch of B windows; K steps are timed from the first mel to the last token, fill and drain included.
        n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count
        enc_cus = args.pipeline if args.pipeline < n_cu else max(1, n_cu * 3 // 8)     # a smaller device: the same 3 : 5 split
        dec_cus = n_cu - enc_cus
        try:
            es = E.Stream(local_rank, 0, enc_cus)
            ds = E.Stream(local_rank, enc_cus, dec_cus)
            dss = [ds]
            full = E.Stream(local_rank, 0, 0)      # fill and drain run alone: every CU
            pst = [st, E.State(ctx, B)]
            pipe = {"encoder_cus": enc_cus, "decoder_cus": dec_cus, "batches_in_flight": 2}
        except E.WhisperError as ex:               # no CU-masked queues on this box: one batch after the other
            print(f"[bench] two-batch pipeline unavailable ({ex}); running one batch after the other", file=sys.stderr, flush=True)
            pipe = None

    if pipe and G > 1:
        pass
    elif pipe:
        def enqueue_front(s_, stream):      # asynchronous: returns as soon as the launches are queued
            s_.set_stream(stream.ptr)
            s_.mel_device(pcm.data_ptr(), pcm.shape[1], n_samples, E.OHW_MEL_ZERO_TAIL)
            s_.encode(B)

        def run_steps(n):           # noqa: F811
            toks = None
            if n > 0:
                enqueue_front(pst[0], full)
                last_front = full
            for i in range(n):
                cur, nxt = pst[i % 2], pst[(i + 1) % 2]
                dstream = ds if i + 1 < n else full
                dstream.wait(last_front)         # batch i's cross-K/V is ready before its decode starts
                if i + 1 < n:
                    es.wait(last_front)          # (the first front end ran on the unrestricted stream)
                    enqueue_front(nxt, es)       # batch i+1's front end runs beside batch i's decode
                    last_front = es
                cur.set_stream(dstream.ptr)
                toks, _ = cur.greedy(B, p)       # host-blocking greedy loop (hipGraph replays on the decoder's stream)
                if use_dist:
                    shard.gather_tokens(shard.pack_tokens(toks, args.tokens), dist, world, rank, dev, force=True)
            return toks

    if pipe and G > 1:
        # set-up, not a step: the decode graphs are captured per (state, decode-batch size) on first use and a capture holds
        # the library's gate against every other lane, so each lane meets every group shape the K timed steps will use before
        # the warm-up (K = 20 on 4 lanes x 3: a group of 12 and one of 8)
        for shape in sorted({min(G * MG, args.steps - g0) for g0 in range(0, args.steps, G * MG)}, reverse=True):
            run_steps(shape)
    toks = run_steps(args.warmup)

    # ---- timed region: exactly K steps, no profiling hooks active (the decode loop replays its hipGraph)
    fence()
    t0 = time.perf_counter()
    toks = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if pipe:
        # the same K steps one batch after the other (untimed for `value`; reported beside it, and the per-stage times
        # below come from this leg, where the stages do not overlap)
        st.set_stream(None)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        pipe["unpipelined_ms_per_step"] = round(1e3 * (time.perf_counter() - t1) / args.steps, 2)
        pipe["unpipelined_value"] = round(30.0 * B * world * args.steps / (time.perf_counter() - t1), 1)   # rank 0's clock
    tm = st.timings()

    # ---- roofline leg (untimed, same workload): HIP events recorded on the launch stream around every
    # launch of one kernel class.  Events cannot be recorded inside a replayed graph, so these passes
    # launch the same kernels eagerly; only per-kernel durations are taken from them, never `value`.
    prof_class = args.profile_class
    class_totals, class_work = {}, {}
    S_saved, S = S, 1          # the roofline leg runs sub-batch 0 alone, eagerly, on its own stream
    def step():                # noqa: F811
        out = [None]
        run_state(0, out)
        return out[0]
    if prof_class == 0:
        for cls in sorted(PROF_NAMES):
            st.profile_begin(cls)
            step()
            n, ms, w = st.profile_end()
            class_totals[cls] = ms
            class_work[cls] = w
        prof_class = max(class_totals, key=class_totals.get)
    st.profile_begin(prof_class)
    for _ in range(args.steps):
        step()
    launches, k_ms, work = st.profile_end()
    # the MFMA class once more at the shape the TIMED schedule launches it: one front-end pass over a lane's MG x B windows (the
    # 32-window figures above pay 8 % for a last round of 256-row tiles that is two thirds empty)
    lane_pass = None
    if pipe and G > 1 and MG > 1:
        s_ = pst[0]
        s_.set_stream(full.ptr)
        lane_pass = {"windows": MG * B}
        for cls, key in ((1, "gemm"), (2, "attention")):
            s_.profile_begin(cls)
            s_.mel_device(pcm_all.data_ptr(), pcm_all.shape[1], n_samples * MG, E.OHW_MEL_ZERO_TAIL)
            s_.encode(MG * B)
            _, ms_c, w_c = s_.profile_end()
            lane_pass[key + "_ms_per_step"] = round(ms_c / MG, 3)
            lane_pass[key + "_tflops"] = round(w_c / (ms_c * 1e-3) / 1e12, 1) if ms_c > 0 else 0.0
        lane_pass["gemm_frac"] = round(lane_pass["gemm_tflops"] / PEAK_MFMA_TFLOPS, 4)

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt_max = float(dt_t.item())
    audio_s = 30.0 * B * world * args.steps
    value = audio_s / dt_max

    assert all(len(t) == args.tokens for t in toks), "every window must decode exactly --tokens tokens"

    if rank == 0:
        if prof_class in (1, 2):
            achieved = work / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_MFMA_TFLOPS, 4), "traffic": None}
        else:
            achieved = work / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None}
        roof["traffic"], roof["traffic_source"] = pmc_traffic(prof_class, B)
        # whole path against its own bounds ("whole-path roofline = sum of stage times at their own bound"):
        # MFMA classes at the dense 16-bit peak, HBM classes (algorithmic bytes) and the mel's 3.46 MB per window at 8 TB/s
        if class_work:
            mfma_ms = 1e3 * (class_work.get(1, 0.0) + class_work.get(2, 0.0)) / (PEAK_MFMA_TFLOPS * 1e12)
            hbm_ms = 1e3 * (sum(class_work.get(c, 0.0) for c in range(3, 9)) + B * (480000 * 4 + hp.n_mels * 3000 * 4)) / (PEAK_HBM_GBS * 1e9)
            ms_step = 1e3 * dt_max / args.steps
            roof["whole_path"] = {"bound_ms_per_step": round(mfma_ms + hbm_ms, 2), "mfma_bound_ms": round(mfma_ms, 2), "hbm_bound_ms": round(hbm_ms, 2),
                                  "measured_ms_per_step": round(ms_step, 2), "frac": round((mfma_ms + hbm_ms) / ms_step, 4)}
        # the path's MFMA-bound class beside the dominant (HBM-bound) kernel: BASELINE.json's north_star asks for both
        mfma = None
        if class_totals.get(1, 0) > 0 and class_work.get(1, 0) > 0:
            tf = class_work[1] / (class_totals[1] * 1e-3) / 1e12
            mfma = {"kernel": PROF_NAMES[1], "bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / PEAK_MFMA_TFLOPS, 4), "ms_per_step": round(class_totals[1], 3),
                    "at_the_timed_schedules_pass": lane_pass,
                    "note": "the chip runs this kernel at its 1.39 kW cap and 2.05 GHz (tools/gemm_power_probe.py): the dense peak at that clock is 2.14 PFLOP/s"}
        roof.update({"kernel": PROF_NAMES[prof_class], "launches": launches, "avg_launch_us": round(1e3 * k_ms / max(1, launches), 2),
                     "kernel_ms_per_step": round(k_ms / args.steps, 3),
                     "class_ms_per_step": {PROF_NAMES[k]: round(v, 3) for k, v in class_totals.items()}})
        latency = None
        if world == 1 and not args.no_latency:
            latency = latency_figures(E, synth, ctx, hp, dtype, args.tokens)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(hp, pcm_host[:max(1, min(args.cpu_windows, B))], args.cpu_tokens or args.tokens)
        print(f"[bench] rank 0 ok: {value:.1f} audio-s/s over {world} rank(s)", file=sys.stderr, flush=True)
        line = {
            "metric": baseline_metric(), "value": round(value, 1), "unit": "audio-sec/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt_max / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "value_one_batch_in_flight": pipe.get("unpipelined_value") if pipe else round(value, 1),
            "config": {"workload": workload_string(args, B, G if pipe and G > 1 else 0, MG, pipe), "model_dims": hp.as_list(),
                       "concurrent_sub_batches": S_saved, "pipeline": pipe,
                       "stage_ms_last_step": {"mel": round(tm.mel_ms, 2), "encode": round(tm.encode_ms, 2), "decode": round(tm.decode_ms, 2)},
                       "decode_steps": tm.decode_steps},
            "roofline": roof, "roofline_mfma": mfma, "latency": latency, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if use_dist:
        # every rank's exit is visible: a rank reports, all ranks meet once more, then tear down (a rank that dies before
        # this line leaves the others' barrier to fail loudly instead of a silent non-zero exit at interpreter shutdown)
        if rank != 0:
            print(f"[bench] rank {rank} ok", file=sys.stderr, flush=True)
        dist.barrier()
        dist.destroy_process_group()
    # release the device objects in a defined order (states and streams before their context) rather than at interpreter exit
    for s_ in states + [x for x in (pipe and pst or []) if x is not st]:
        s_.close()
    if pipe:
        for x in [es, full] + dss:
            x.close()
    ctx.close()


def workload_string(args, B, lanes, merge, pipe):
    """what the timed region ran, in one string (the driver keeps config.workload and drops the nested objects)"""
    base = (f"{args.model} dims, steps of batch={B} x 30 s windows per GPU, greedy, {args.tokens} tokens/window (EOT suppressed), "
            f"procedural weights seed 1234, every window a distinct synthetic recording, PCM resident in HBM")
    if lanes:
        strict = pipe.get("unpipelined_value")
        return (base + f"; schedule LANES {lanes}x{merge}: {lanes * merge} steps ({lanes * merge * B} windows) in flight per GPU - per lane ONE front-end pass "
                f"and ONE decode batch of {merge * B} rows, {lanes} decodes side by side on {pipe['decoder_cus']} CUs each; the same steps strictly one "
                f"batch of {B} in flight: {strict} audio-s/s (value_one_batch_in_flight)")
    if pipe:
        return base + f"; schedule PIPELINE: 2 batches in flight (front end on {pipe['encoder_cus']} CUs beside the decode on {pipe['decoder_cus']})"
    return base + "; one batch after the other"


def decoder_step_bytes(hp, n_rows_windows=1):
    """algorithmic HBM bytes of ONE single-token decoder step: every decoder weight once (16-bit) + the cross
    K/V of every window once; the self-attention cache (a few hundred KB per window) is left out"""
    d, L, V, T = hp.n_text_state, hp.n_text_layer, hp.n_vocab, hp.n_audio_ctx
    weights = (L * 14 * d * d + V * d) * 2
    xkv = n_rows_windows * L * 2 * T * d * 2
    return weights + xkv


def latency_figures(E, synth, ctx, hp, dtype, n_tokens):
    """The small-batch figures (untimed for `value`): what the daemon's one-utterance path and BASELINE configs #2 / #5 see.
    Each is the median of 3 runs after one warm-up, host wall clock around the blocking C-ABI calls."""
    import statistics as stat
    out = {}

    def med(f, n=3):
        f()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); f(); ts.append(1e3 * (time.perf_counter() - t0))
        return stat.median(ts)
    pcm1 = synth.synth_audio(777)[None]
    # (1) large-v3 (the bench's model) at batch 1: greedy decode, ms per token against the step's byte bound
    st = E.State(ctx, 1)
    p = ctx.default_params(); p.force_len = n_tokens
    st.mel(pcm1, None, E.OHW_MEL_ZERO_TAIL, want=False); st.encode(1)
    dec_ms = med(lambda: st.greedy(1, p))
    front_ms = med(lambda: (st.mel(pcm1, None, E.OHW_MEL_ZERO_TAIL, want=False), st.encode(1), st.fetch("mel", 1)))
    per_tok = dec_ms / n_tokens
    bound_ms = 1e3 * decoder_step_bytes(hp) / (PEAK_HBM_GBS * 1e9)
    out["b1_large_v3_ms_per_token"] = {"value": round(per_tok, 4), "hbm_bound_ms": round(bound_ms, 4), "frac": round(bound_ms / per_tok, 4),
                                       "window_ms": round(front_ms + dec_ms, 2), "front_end_ms": round(front_ms, 2), "tokens": n_tokens}
    # the same decode through the one-launch persistent step (decode_persist.hip; off by default: its all-to-all hand-offs cost
    # more than the kernel boundaries they replace) - recorded so that the rejected design has a driver-run number beside it
    st.set_persistent(True)
    out["b1_large_v3_ms_per_token"]["persistent_step_ms_per_token"] = round(med(lambda: st.greedy(1, p)) / n_tokens, 4)
    st.close()
    # (2) BASELINE config #5: one 5 s chunk, beam = 5, 48 decoder steps (procedural weights never finish early), mel + encoder +
    # cross K/V + beam search on one window (tools/streaming_latency.py drives the same through StreamingSession)
    K, steps5 = 5, 48
    st5 = E.State(ctx, K)
    p5 = ctx.default_params(); p5.n_max = steps5
    chunk = synth.synth_audio(778)[:80000][None]

    def cfg5():
        st5.mel(chunk, [80000], E.OHW_MEL_ZERO_TAIL, want=False); st5.encode(1); st5.beam_search(1, K, p5)
    c5 = med(cfg5)
    b5 = 1e3 * steps5 * decoder_step_bytes(hp) / (PEAK_HBM_GBS * 1e9)
    out["cfg5_beam5_chunk_ms"] = {"value": round(c5, 2), "hbm_bound_ms_decode": round(b5, 2), "frac": round(b5 / c5, 4), "beam": K, "decoder_steps": steps5,
                                  "chunk_s": 5.0}
    st5.close()
    # (3) BASELINE config #2: `small` dims, batch 1, one 30 s window, greedy, the same token count
    hs = synth.PRESETS["small"]
    cs = E.Context.synthetic(hs.as_list(), 1234, 0, dtype)
    ss = E.State(cs, 1)
    ps = cs.default_params(); ps.force_len = n_tokens

    def cfg2():
        ss.mel(pcm1, None, E.OHW_MEL_ZERO_TAIL, want=False); ss.encode(1); ss.greedy(1, ps)
    c2 = med(cfg2)
    enc_flop = 2.0 * (1500 * (3 * 80 * hs.n_audio_state) * 2 + 1500 * 3 * hs.n_audio_state ** 2) + hs.n_audio_layer * (
        2.0 * 1500 * 12 * hs.n_audio_state ** 2 + 4.0 * 1500 * 1500 * hs.n_audio_state) + 2.0 * 1500 * 2 * hs.n_text_layer * hs.n_text_state ** 2
    b2 = 1e3 * (enc_flop / (PEAK_MFMA_TFLOPS * 1e12) + n_tokens * decoder_step_bytes(hs) / (PEAK_HBM_GBS * 1e9))
    out["cfg2_small_b1_ms_per_window"] = {"value": round(c2, 2), "bound_ms": round(b2, 3), "frac": round(b2 / c2, 4), "tokens": n_tokens}
    ss.close(); cs.close()
    return out


def pool_bench(args, torch, E, synth):
    """bench.py --pool: the single-process multi-GPU design behind the C ABI (openhush_amd/csrc/pool.cpp).  A step is still one
    batch of --batch windows per GPU; the K timed steps are ONE ohw_pool_transcribe of a recording of gpus x batch x K windows
    (host PCM in, token ids out: PCIe-inclusive, unlike the default mode), every window forced to --tokens tokens."""
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--pool is ONE process: launch it plainly, not under torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    devices = [int(x) for x in args.pool_devices.split(",")] if args.pool_devices else list(range(args.gpus))
    n_dev = len(devices)
    hp = synth.PRESETS[args.model]
    dtype = E.OHW_DTYPE_BF16 if args.dtype == "bf16" else E.OHW_DTYPE_F16
    B = args.batch
    t0 = time.perf_counter()
    pool = E.EnginePool(None, "en", False, devices, dtype, B, synthetic=hp.as_list(), seed=1234)
    load_s = time.perf_counter() - t0
    pool.set_decode_policy(temperature_inc=0.0)
    pool.set_force_len(args.tokens)
    if args.phases > 1:
        pool.set_schedule(E.OHW_SCHEDULE_LANES, args.phases, max(1, args.merge))
    else:
        pool.set_schedule(E.OHW_SCHEDULE_SEQUENTIAL if args.pipeline <= 0 else E.OHW_SCHEDULE_PIPELINE)

    # ohw_pool_transcribe has ohw_engine_transcribe's contract, validate_audio included: at most 2 h per call (the reference's limit,
    # src/engine/validation.rs:67-72) = 240 windows.  The K steps are as many calls as that takes, windows distinct throughout.
    MAX_WIN = 240

    def recording(n_win, first_id):
        out = np.empty(n_win * synth.CHUNK_SAMPLES, np.float32)
        for w in range(n_win):
            out[w * synth.CHUNK_SAMPLES:(w + 1) * synth.CHUNK_SAMPLES] = synth.synth_audio(first_id + w)
        return out

    def calls(n_steps, first_id):
        left, at, recs = n_dev * B * n_steps, first_id, []
        while left > 0:
            n = min(left, MAX_WIN)
            recs.append(recording(n, at))
            at += n; left -= n
        return recs
    for r in calls(args.warmup, 0):
        pool.transcribe(E.AudioBuffer(r, 16000))
    recs = calls(args.steps, 100000)
    torch.cuda.synchronize()
    lens = []
    t0 = time.perf_counter()
    for r in recs:
        pool.transcribe(E.AudioBuffer(r, 16000))
        lens += pool.last_window_tokens()
    dt = time.perf_counter() - t0
    assert len(lens) == n_dev * B * args.steps and all(n == args.tokens for n in lens), "every window must decode exactly --tokens tokens"
    value = 30.0 * len(lens) / dt
    line = {
        "metric": baseline_metric(), "value": round(value, 1), "unit": "audio-sec/sec", "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"--pool: ONE process, ohw_pool_transcribe over devices {devices} ({pool.broadcast_kind} weight broadcast"
                               f"{': ' + pool.broadcast_note if pool.broadcast_note else ''}), {args.model} dims, {len(recs)} recording(s) of at most 240 windows (2 h, the validation limit of the call), {len(lens)} distinct 30 s windows = "
                               f"{args.steps} steps of batch={B} per GPU dealt round-robin, greedy, {args.tokens} tokens/window (EOT suppressed), HOST PCM in "
                               f"(PCIe-inclusive), engine schedule {'LANES %dx%d' % (args.phases, args.merge) if args.phases > 1 else 'sequential/pipeline'}",
                   "model_dims": hp.as_list(), "pool_load_s": round(load_s, 2), "broadcast": pool.broadcast_kind},
        "roofline": None, "cpu_baseline": None,
    }
    print(json.dumps(line), flush=True)
    pool.close()


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file sits beside bench.py; the fallback is its text at round 1)"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, ValueError, KeyError):
        return "audio-sec/sec (xRT) large-v3 greedy, 30s chunks, 1/2/4/8 MI355X"


def pmc_traffic(prof_class, batch):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from).  bench.py cannot run the profiler
    around itself: the figure is read from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_summary.json:
    FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, separate --pmc passes of this command at a reduced token
    count, tools/run_profiles.sh) and labelled as such; (None, reason) when no committed figure matches this workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            with open(path) as f:
                pm = json.load(f)
            ent = pm.get(str(prof_class))
            if ent and ent.get("batch") == batch:
                return ent["hbm_bytes_per_launch"], (f"committed PMC pass profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                                     "separate runs of this command at --tokens 6), not measured in this run")
        except (OSError, ValueError, KeyError):
            continue
    return None, "no committed PMC pass matches this kernel class and batch"


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a box that
    shows 128 logical CPUs but grants 16 would otherwise run 128 spinning OpenMP threads on 16 cores)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, int(os.environ.get("OHW_CPU_BASELINE_THREADS", "16"))))


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(hp, pcm, n_tokens):
    """The oracle (kind "port") on this host's cores, a bounded sample of the same workload, at the two thread counts
    SURVEY.md 8d names: 4 (the reference engine's effective default: whisper.cpp n_threads = min(4, cores)) and every
    usable core.  `value` / `cores` are the all-cores leg; `runs` holds both."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # before libgomp loads: no spinning if oversubscribed
    from oracle import oracle
    cores = usable_cores()
    oracle.set_num_threads(cores)
    print(f"[bench] cpu_baseline: generating the model on the host ({cores} threads) ...", file=sys.stderr, flush=True)
    m = oracle.Model.synth(hp.as_list(), 1234)
    p = m.default_params()
    p.force_len = n_tokens
    runs = []
    legs = [(cores, len(pcm))] + ([(4, 1)] if cores > 4 else [])       # the 4-thread leg: one window (about 30 s of CPU work)
    for threads, n_win in legs:
        oracle.set_num_threads(threads)
        t0 = time.perf_counter()
        t_mel = t_enc = t_dec = 0.0
        for w in range(n_win):
            toks, (a, b, c) = m.transcribe_chunk(pcm[w], p, 1)
            assert len(toks) == n_tokens
            t_mel += a; t_enc += b; t_dec += c
            print(f"[bench] cpu_baseline: {threads} threads, window {w + 1}/{n_win} done", file=sys.stderr, flush=True)
        dt = time.perf_counter() - t0
        runs.append({"cores": threads, "value": round(30.0 * n_win / dt, 3), "windows": n_win,
                     "stage_s": {"mel": round(t_mel, 2), "encoder+crossKV": round(t_enc, 2), "decode": round(t_dec, 2)}})
    m.close()
    top = runs[0]
    return {"value": top["value"], "unit": "audio-sec/sec", "cores": top["cores"], "kind": "port", "cpu_model": cpu_model_name(),
            "sample": f"{top['windows']} windows (30 s each) of the same workload, {n_tokens} decode tokens per window, fp32 C/OpenMP oracle "
                      f"(x86-64-v3): mel {top['stage_s']['mel']:.2f} s, encoder+crossKV {top['stage_s']['encoder+crossKV']:.2f} s, "
                      f"decode {top['stage_s']['decode']:.2f} s; second leg at 4 threads (the reference engine's default) on 1 window",
            "runs": runs}


if __name__ == "__main__":
    main()
