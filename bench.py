#!/usr/bin/env python3
"""Headline benchmark: ResNet-50 W8A8 per-channel fake-quant forward (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W [--scaling strong] [--model resnet50|repvgg_a1|mobileone_s1|resnet18]
    N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or on its own - without
    WORLD_SIZE in the environment `python bench.py --gpus N` starts that launcher itself as a child process BEFORE anything
    touches a GPU (the reference spawns its ranks the same way, example/quantization/DDP_RootQ_train.py:30-34), forwards rank 0's
    JSON line and exits with the child's status.
    BASELINE configs[3] verbatim: --model repvgg_a1 --scaling strong --global-batch 4096 --gpus 8;
    configs[4]: --model mobileone_s1 (W4A8 asymmetric per-channel weights, QBase family, 1024 images per GPU).

A step = one forward of the quantised ResNet-50 (54 quantised layers, FSPTQ forms: weights minmax_channel s8, activations
minmax_tensor u8, BatchNorm folded first as FSPTQuant.py:67 does) over 512 synthetic 224x224 images per GPU, scales frozen
(the observer pass and its all-reduce run once, before the timed region).  Default mode = the frozen execution plan of
dlmc/utils/fuse.py: every layer is ONE launch of an int8 MFMA kernel (csrc/conv_i8.hip) whose epilogue dequantises, adds
the shortcut, applies ReLU and emits the NEXT layer's activation codes, and 11 block boundaries run their two layers as one
launch (csrc/conv_chain_i8.hip); weight codes are quantised once, when the plan is
built (the reference re-quantises 0.2 GB of weights per forward: 0.5 % of the step's bytes).  `--plan modules` and
`--conv fp32` run the reference's own op sequence (stand-alone fake-quant kernels + MIOpen convolutions).
Inputs are resident in HBM when the timed region starts.  Weak scaling by default (512 images on every GPU, no collective
in the steady state); `--scaling strong` splits a fixed global batch of 4096.

One JSON line on rank 0.  Besides the contract fields:
  roofline             the kernel with the largest share of the step - since round 2 the chain kernel (a block's last 1x1
                       convolution + shortcut + ReLU + quantiser + the next block's first 1x1, csrc/conv_chain_i8.hip):
                       algorithmic bytes / HIP-event durations of its launches in the first timed step, on the launch stream;
                       `traffic` comes from the committed rocprofv3 PMC passes of the same command (`traffic_measured_in_run`
                       says so: counters cannot be read from inside the process)
  roofline_second_kernel  the same for the other int8 kernel (csrc/conv_i8.hip: 3x3s, stage 3 / 4 block ends, first 1x1s, fc)
  first_batch          one more calibrating forward (every observer re-armed): SURVEY 8(d)'s "first batch, observer on", through
                       dlmc.utils.fuse.EagerFused (layer + shortcut add + ReLU as one int8 launch; same scales); `module_path_ms`
                       = the same through the wrappers alone (torch's ReLU / add between them; the one host read per layer - is
                       the zero point an integer? - costs 1.7 of its ~30 ms, tools/first_batch_probe.py)
  roofline_fake_quant  the stand-alone fake-quant kernel, measured live on BASELINE configs[1]'s tensor
  quant_work_8d        SURVEY.md 8(d)'s own accounting of the step: the bytes the reference's fake-quant passes move for this
                       batch (8 B per fake-quantised activation / weight element) over the WHOLE step's time, against 8 TB/s.
                       An equivalent rate, not HBM traffic: the fused plan never moves most of those bytes.
  roofline_third_kernel   (round 4) the halo-tile 3x3 kernel and the generic kernel are separate families now (`conv3x3_halo` / `conv_i8`):
                       second = the one with the larger share of the step, third = the other; the weight-resident block-end kernel
                       (`conv_pwr`) joins the ranking: the step's fourth family is `roofline_fourth_kernel`
  other_configs        (default model only) BASELINE configs[3]'s network (RepVGG-A1, 512 images per GPU) and configs[4] (MobileOne-S1
                       W4A8, 1024 per GPU) through the same pipeline, a short run each AFTER the headline region: value, ms_per_step,
                       steps and the roofline object of the family with the largest share of a profiled step
  collective           (N > 1) what the collective library saw: backend, world, ranks counted by an all-reduce of ones, RCCL version,
                       the number of observer all-reduces the calibrating forward issued and the time of as many synchronous
                       all_reduce(MAX) calls through the observers' own function; the steady state has no collective
  cpu_baseline         the CPU port of the same layer stack (oracle/), timed on this box's host cores on a bounded sample
                       (N = 1 only): all granted cores and one thread, median and min
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dlmc-quant_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 measured for a float4 copy
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8 on the matrix cores (2x the bf16 rate; guides/MI355X_MICROARCH.md)

QCFG = {  # example/quantization/FSPTQ_config.yaml:40-53: W minmax_channel s8, A minmax_tensor u8
    "weight": {"enable": True, "type": "minmax_channel", "recon_type": "None", "args": {"n_bits": 8, "signed": True}},
    "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
    "exclude_layers": [], "override_options": [],
}


def usable_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box
    shows 256 logical CPUs but grants a share of them; oversubscribing torch's pool is catastrophic)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model_name():
    """The host CPU's model string (BASELINE.md section 3 promises it beside the core count)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(batch, budget_s=16.0, fold_bn=True, halfnormal=True):
    """The reference's CPU path (port in oracle/ref_layers.py) on a bounded sample of the same workload: all granted
    cores (the headline `value`, from the median forward) and one thread, median and min of the timed forwards."""
    import workloads as W
    from oracle.ref_layers import port_model
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(2333)
    model = W.resnet50().eval()
    if fold_bn:
        from oracle import fakequant_oracle as O
        for name, m in list(model.named_modules()):      # the same BN folding, by the oracle, on the CPU model
            for cname, child in list(m.named_children()):
                if isinstance(child, torch.nn.BatchNorm2d):
                    target = cname.replace("bn", "conv") if not cname.isdecimal() else str(int(cname) - 1)
                    conv = getattr(m, target)
                    w, b = O.fold_bn(conv.weight.data, None if conv.bias is None else conv.bias.data, child.weight.data,
                                     child.bias.data, child.running_mean, child.running_var)
                    conv.weight.data, conv.bias = w, torch.nn.Parameter(b)
                    setattr(m, cname, torch.nn.Identity())
    model = port_model(model, "FSPTQ")
    x = torch.randn(batch, 3, 224, 224)
    if halfnormal:
        x = torch.relu(x)

    def timed(budget, max_reps):
        ts = []
        while not ts or (sum(ts) + sum(ts) / len(ts) < budget and len(ts) < max_reps):
            t0 = time.perf_counter()
            model(x)
            ts.append(time.perf_counter() - t0)
        return ts
    with torch.no_grad():
        model(x)                      # calibration + warm-up
        ts = timed(budget_s, 50)
        torch.set_num_threads(1)
        t1 = timed(budget_s / 2, 5)
        torch.set_num_threads(cores)
    med, med1 = statistics.median(ts), statistics.median(t1)
    return {"value": round(batch / med, 2), "unit": "images/s", "cores": cores, "cpu_model": cpu_model_name(), "kind": "port",
            "sample": f"{len(ts)} forwards of batch {batch} (same ResNet-50 W8A8 layer stack, torch {torch.__version__} CPU, "
                      f"{cores} threads), after 1 calibration forward; value = batch / median forward time",
            "median_ms": round(med * 1e3, 1), "min_ms": round(min(ts) * 1e3, 1),
            "one_thread": {"value": round(batch / med1, 2), "median_ms": round(med1 * 1e3, 1), "min_ms": round(min(t1) * 1e3, 1),
                           "forwards": len(t1)}}


W4A8 = {  # BASELINE configs[4]: QBase family (quantization_type=None), asymmetric per-channel u4 weights (ops.py:129-136), u8 activations
    "weight": {"enable": True, "type": "minmax_channel", "args": {"n_bits": 4, "signed": False}},
    "input": {"enable": True, "type": "minmax_tensor", "args": {"n_bits": 8, "signed": False}},
    "exclude_layers": [], "override_options": [],
}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run as a CHILD process (this
    process has not touched a GPU and never will), pass rank 0's JSON line through, return the child's exit status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                 # rank 0 prints exactly one JSON line; anything else is passed to stderr
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    return rc if rc != 0 or line is not None else 1


def dry_run(args, world, rank):
    """--dry-run (CPU, no GPU work; tests/test_bench_launcher.py): the launcher, the rendezvous, the barrier-bracketed timed
    region, the MAX over ranks and the one JSON line - with an empty step.  The line says so and carries no measurement."""
    coll = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        coll = collective_report(torch.device("cpu"), world, args.backend, 0)     # the same object a real N > 1 line carries
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": f"{args.model} fake-quant fwd images/sec", "value": 0.0, "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.0, "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "dtype": "i8", "data": "dry-run: no GPU work, no measurement",
                          "config": {"workload": "dry run of the launcher and the rendezvous", "global_batch": args.batch * world,
                                     "parallelism": f"dp{world} (batch-sharded replicas)", "backend": args.backend},
                          **({"collective": coll} if coll else {})}), flush=True)


def families(records):
    """HIP-event durations of the profiled launches by kernel family (tag): launches, algorithmic bytes, integer operations, ms."""
    fam = {}
    for tag, nbytes, a, b, ops in records:
        f = fam.setdefault(tag, {"launches": 0, "bytes": 0, "ms": 0.0, "ops": 0})
        f["launches"] += 1
        f["bytes"] += nbytes
        f["ops"] += ops
        f["ms"] += a.elapsed_time(b)
    return fam


FAMILY_KERNEL = {"conv_chain": "conv_chain_i8_kernel", "conv_i8": "conv_i8_mfma_kernel", "conv3x3_halo": "conv3x3_halo_i8_kernel",
                 "conv_dw": "conv_dw3_i8_kernel / conv_dw3p2_i8_kernel", "conv_dwm": "conv_dwm_i8_kernel", "conv_stem": "conv_stem_i8_kernel / conv_stem_pool7_i8_kernel",
                 "conv_dwpw": "conv_dwpw_i8_kernel", "conv_pw": "conv_pw_i8_kernel", "conv_pwr": "conv_pwr_i8_kernel", "fq_image": "quantize_pad_nhwc4_kernel"}


def family_roofline(tag, f):
    """The roofline object of one kernel family of a profiled step: the bound is the roofline that needs more time at its peak."""
    sec = f["ms"] * 1e-3
    gbps, tops = f["bytes"] / sec / 1e9, f["ops"] / sec / 1e12
    t_hbm, t_mfma = f["bytes"] / (HBM_PEAK_GBPS * 1e9), f["ops"] / (MFMA_I8_PEAK_TOPS * 1e12)
    r = {"kernel": FAMILY_KERNEL.get(tag, tag), "launches": f["launches"], "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2),
         "hbm": {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4)},
         "mfma": {"achieved": round(tops, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops / MFMA_I8_PEAK_TOPS, 4)}}
    b = "mfma" if t_mfma > t_hbm else "hbm"
    return {"bound": b, "achieved": r[b]["achieved"], "peak": r[b]["peak"], "unit": r[b]["unit"], "frac": r[b]["frac"], "traffic": None, **r}


def pmc_traffic(kernel_substr, launches, algorithmic, pattern="*pmc_bench_fused_plan*.json", directory=None):
    """HBM bytes per launch of a kernel family from the committed PMC passes of this same command, run with --no-other-configs
    (profiles/*pmc_bench*.json, made by tools/pmc_summary.py --tail; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    16-byte-per-lane streaming reads on gfx950).  A TAIL entry is used only if it describes THIS step's launches: its dispatch
    count must equal the family's launches per step, and its bytes must not be below 0.9 x the family's algorithmic bytes (round 4's
    file held the side networks' dispatches for two families: less traffic than the layers' own operands - refused now, with the reason).
    Only the NEWEST file (highest round) that knows the kernel is asked: an older round's passes describe an older build.
    Returns (bytes, file, reason-if-refused)."""
    import glob
    why = f"no profiles/{pattern} with a TAIL entry for this kernel"
    directory = directory or os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for path in sorted(glob.glob(os.path.join(directory, pattern)), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        known = False
        for k, v in d.items():          # (a file may hold several tails of a kernel - different dispatch counts: the step's is the one that counts)
            if k.startswith("TAIL") and kernel_substr in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                known = True
                got = int((2 * v["FETCH_SIZE"]["mean_KiB"] + v["WRITE_SIZE"]["mean_KiB"]) * 1024)
                n = (v["FETCH_SIZE"]["dispatches"], v["WRITE_SIZE"]["dispatches"])
                if n != (launches, launches):
                    if "below 0.9" not in why:
                        why = f"{os.path.basename(path)}: TAIL holds {n} dispatches, the step has {launches} launches of this kernel"
                elif algorithmic and got < 0.9 * algorithmic:
                    why = (f"{os.path.basename(path)}: TAIL reads {got} B per launch, below 0.9 x the {algorithmic} algorithmic bytes - "
                           "not this step's launches; not reported")
                else:
                    return got, os.path.basename(path), None
        if known:
            return None, None, why
    return None, None, why


def side_roofline(name, tag, f):
    """A side network's roofline object with the HBM traffic of ITS launches (profiles/*pmc_bench_<name>.json: rocprofv3 passes of
    `bench.py --model <name>`, tools/collect_profiles.sh), under the same plausibility rule as the headline's."""
    r = family_roofline(tag, f)
    kern = FAMILY_KERNEL.get(tag, tag).split(" ")[0]
    got, src, why = pmc_traffic(kern, f["launches"], f["bytes"] // max(f["launches"], 1), f"*pmc_bench_{name}*.json")
    r["traffic"] = got
    r["traffic_measured_in_run"] = False
    if src:
        r["traffic_source"] = f"profiles/{src}"
        r["traffic_over_algorithmic"] = round(got / max(f["bytes"] // max(f["launches"], 1), 1), 4)
    else:
        r["traffic_unavailable"] = why
    return r


def side_config(name, batch, args, dev, world, rank, barrier):
    """One more network of BASELINE.json through the same pipeline as the headline (calibrating forward, frozen plan, K steps
    bracketed by barrier + synchronize, MAX over ranks): repvgg_a1 = configs[3]'s network at its 512 images per GPU (FSPTQ W8A8,
    deploy form), mobileone_s1 = configs[4] (W4A8 asymmetric per-channel weights, 1024 images per GPU).  One extra step in front
    of the timed region carries the per-kernel HIP events (one stream); `roofline` is the family with the largest share of it."""
    import workloads as W
    from dlmc.quantization.scalar import kernels as K
    from dlmc.utils.fuse import StreamedPlan, fuse_inference
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    model = W.MODELS[name]().to(dev).eval()
    w4a8 = name == "mobileone_s1"
    if w4a8:
        quantize_model(model, json.loads(json.dumps(W4A8)), None, int8_gemm=True)
    else:
        from dlmc.utils.merge_bn import merge_bn
        model = merge_bn(model, inplace=True, allow_missing=True)
        quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=True)
    g = torch.Generator(device=dev).manual_seed(2333 + rank)
    x = torch.relu(torch.randn(batch, 3, 224, 224, device=dev, generator=g))
    steps, warmup = args.other_steps, 10
    with torch.no_grad():
        model(x)                                      # calibration (observers + their all-reduce), untimed
        single = fuse_inference(model)
        plan = StreamedPlan(single, args.streams) if args.streams > 1 and not w4a8 else single
        for _ in range(warmup):
            plan(x)
        single(x)
        K.PROFILE.reset()
        K.PROFILE.enabled = True                      # one more single-stream step with per-kernel HIP events, OUTSIDE the timed region
        single(x)
        K.PROFILE.enabled = False
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            plan(x)
        barrier()
        elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    fam = {k: f for k, f in families(K.PROFILE.records).items() if f["ms"] > 0}
    K.PROFILE.reset()
    top = max(fam, key=lambda k: fam[k]["ms"])
    return {"metric": f"{name} {'W4A8' if w4a8 else 'W8A8'} fake-quant fwd images/sec", "value": round(batch * world * steps / elapsed, 1),
            "unit": "images/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup, "batch_per_gpu": batch,
            "config": ("BASELINE configs[4]: W minmax_channel u4 asymmetric per channel (QBase family), A minmax_tensor u8, packed int4 weights"
                       if w4a8 else "BASELINE configs[3]'s network at its per-GPU batch: FSPTQ W minmax_channel s8, A minmax_tensor u8, deploy form"),
            "roofline": side_roofline(name, top, fam[top]),
            "families_ms": {k: round(f["ms"], 3) for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}}


def fake_quant_leg(args, dev, x, steps=3):
    """north_star's literal bar on the headline model: the SAME network with every wrapper on its own (`--conv fp32 --plan modules`:
    per layer one stand-alone fake-quant launch for the activations and one for the weights - fq_tensor_kernel / fq_channel_kernel, 4 B read
    + 4 B written per element: SURVEY.md 8(d)'s 43.9 GB per step at batch 512 - then MIOpen's fp32 convolution of the fake-quantised
    operands, the reference's own op sequence), `steps` forwards with HIP events on every fake-quant launch after one calibrating
    forward.  The roofline object is algorithmic bytes / summed launch time of those kernels alone; the convolutions in between are not in it."""
    import workloads as W
    from dlmc.quantization.scalar import kernels as K
    from dlmc.utils.quantize import quantize_model
    torch.manual_seed(2333)
    model = W.MODELS[args.model]().to(dev).eval()
    if not args.keep_bn:
        from dlmc.utils.merge_bn import merge_bn
        model = merge_bn(model, inplace=True, allow_missing=True)
    quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=False)
    with torch.no_grad():
        model(x)                     # calibration, untimed
        model(x)                     # warm-up (MIOpen's algorithm search)
        torch.cuda.synchronize()
        K.PROFILE.reset()
        K.PROFILE.enabled = True
        t0 = time.perf_counter()
        for _ in range(steps):
            model(x)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        K.PROFILE.enabled = False
    fam = {k: f for k, f in families(K.PROFILE.records).items() if k.startswith("fq") and f["ms"] > 0}
    K.PROFILE.reset()
    nbytes, ms, launches = sum(f["bytes"] for f in fam.values()), sum(f["ms"] for f in fam.values()), sum(f["launches"] for f in fam.values())
    gbps = nbytes / (ms * 1e-3) / 1e9 if ms else 0.0
    del model
    torch.cuda.empty_cache()
    return {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": None,
            "kernel": "fq_tensor_kernel<ZEROPOINT> + fq_channel_kernel<SYMMETRIC> (stand-alone fake-quant of every layer's input and weight)",
            "launches_per_step": launches // steps, "algorithmic_bytes_per_step": nbytes // steps, "fake_quant_ms_per_step": round(ms / steps, 3),
            "whole_step_ms": round(wall / steps * 1e3, 2), "steps": steps,
            "families": {k: {"launches": f["launches"] // steps, "GBps": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1)} for k, f in fam.items()},
            "note": f"{args.model} batch {args.batch} in `--conv fp32 --plan modules` form (the reference's op sequence: fake-quant launches + fp32 "
                    "convolutions), same process, after the headline region; bytes = 8 per fake-quantised element (SURVEY.md 8(d): 43.9 GB per "
                    "step for ResNet-50 at batch 512), time = the fake-quant launches' HIP events summed; north_star's >= 0.70 is about this figure"}


def collective_report(dev, world, backend, calib_allreduces, reps=54):
    """What the collective library saw (N > 1): ranks counted by an all-reduce of ones, the library's version, and the cost of the
    calibrating forward's exchange step - `reps` synchronous all_reduce(MAX) of a packed [max | -min] pair through the very function
    the observers call (dlmc/quantization/scalar/_wrapper.py: allreduce_minmax; ResNet-50 has 54 activation observers)."""
    from dlmc.quantization.scalar._wrapper import allreduce_minmax
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    v, m = torch.rand(1, device=dev), torch.rand(1, device=dev)
    allreduce_minmax(v, m)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        v, m = allreduce_minmax(v, m)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(i) for i in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001  (a version query must not fail the run)
            ver = f"unavailable ({e})"
    return {"backend": backend, "library": "RCCL (torch.distributed 'nccl' on ROCm)" if backend == "nccl" else backend, "world": world,
            "ranks_seen": int(round(float(ones.item()))), "rccl_version": ver, "observer_allreduces_in_calibration": calib_allreduces,
            "observer_allreduce_ms": round(ms, 3), "observer_allreduce_calls_timed": reps,
            "steady_state_collectives_per_step": 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (weak scaling); default 512, mobileone_s1: 1024")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch images on every GPU; strong: a fixed global batch (--global-batch, default 4096 = "
                         "BASELINE configs[3]'s 8 x 512) split evenly over the ranks")
    ap.add_argument("--global-batch", type=int, default=4096, help="--scaling strong: images per step over all GPUs")
    ap.add_argument("--model", default="resnet50", choices=["resnet18", "resnet50", "repvgg_a1", "mobileone_s1"],
                    help="resnet50 = BASELINE configs[2] (the headline); repvgg_a1 = configs[3]'s network (FSPTQ W8A8, deploy form); "
                         "mobileone_s1 = configs[4] (W4A8, asymmetric per-channel weights, QBase family)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL over xGMI)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: launcher + rendezvous + one (empty) JSON line; CPU tests")
    ap.add_argument("--keep-bn", action="store_true",
                    help="skip the BN folding of the reference's few-shot PTQ flow (FSPTQuant.py:67) and keep BatchNorm layers")
    ap.add_argument("--conv", choices=["int8", "fp32"], default="int8",
                    help="int8 (default): fused int8-dequant x GEMM convs/linears on the matrix cores - BASELINE configs[2] names "
                         "this path; fp32: MIOpen fp32 conv of the fake-quantised operands (the reference's own op sequence)")
    ap.add_argument("--plan", choices=["fused", "modules"], default="fused",
                    help="fused (default, int8 only): the frozen execution plan of dlmc.utils.fuse - ReLU, residual add and the "
                         "next layer's activation quantiser folded into the int8 kernel's epilogue (bit-identical results); "
                         "modules: every wrapper runs on its own, as the reference's module graph does")
    ap.add_argument("--input", choices=["halfnormal", "normal"], default="halfnormal",
                    help="synthetic pixels: relu(N(0,1)) (default; SURVEY.md 8(d): the input of unsigned-activation configs, "
                         "min = 0 -> integer zero point) or N(0,1) (the u8 zero point of the first layer is then the negative "
                         "non-integer minimum, as the reference computes it, and that layer keeps its fp32 path)")
    ap.add_argument("--streams", type=int, default=2,
                    help="fused plan only: split each step's batch over this many HIP streams (dlmc.utils.fuse.StreamedPlan, "
                         "bit-identical results; 2 gives +5-9 %% on ResNet-50 b512: the ramp and tail of one layer's launch overlap "
                         "the other half-batch's neighbouring layer).  The steps that carry per-kernel HIP events "
                         "(--profiled-steps, the first of the timed region) run on ONE stream, so that the roofline objects "
                         "describe kernels that had the chip to themselves, as the rocprofv3 summaries do")
    ap.add_argument("--profiled-steps", type=int, default=1,
                    help="timed steps whose launches carry HIP events (per-kernel durations for the roofline objects)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds of CPU work for the all-cores leg of cpu_baseline (half of it again on one thread)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="default model only: skip the short RepVGG-A1 b512 / MobileOne-S1 b1024 runs reported under `other_configs`")
    ap.add_argument("--other-steps", type=int, default=50, help="timed steps of each `other_configs` run (10 warm-up steps)")
    ap.add_argument("--row-major-blocks", action="store_true",
                    help="A/B: keep the fp32 block tensors between chain kernels row-major (fuse_inference(block_layout=False)) instead of chunk-major")
    ap.add_argument("--no-fake-quant-leg", dest="fq_leg", action="store_false",
                    help="skip `roofline_fake_quant_resnet50`: three forwards of the same model in `--conv fp32 --plan modules` form behind the "
                         "headline region (the stand-alone fake-quant kernels' own roofline on the whole network)")
    args = ap.parse_args()
    args.int8 = args.conv == "int8"
    if args.batch is None:
        args.batch = 1024 if args.model == "mobileone_s1" else 512
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # (nothing has touched a GPU: importing torch does not)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.scaling == "strong":
        if args.global_batch % world:
            sys.exit("bench.py: --global-batch must be a multiple of the number of GPUs")
        args.batch = args.global_batch // world
    if args.dry_run:
        return dry_run(args, world, rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import workloads as W
    from dlmc.quantization.scalar import kernels as K
    from dlmc.utils.quantize import quantize_model

    torch.manual_seed(2333)  # the reference's seed; identical weights on every rank (replicated)
    model = W.MODELS[args.model]().to(dev).eval()
    w4a8 = args.model == "mobileone_s1"
    if w4a8:       # deploy form (no BatchNorm); the W4A8 per-channel configuration runs through the QBase family
        quantize_model(model, json.loads(json.dumps(W4A8)), None, int8_gemm=args.int8)
    else:
        if not args.keep_bn:
            from dlmc.utils.merge_bn import merge_bn
            model = merge_bn(model, inplace=True, allow_missing=True)   # FSPTQuant.py:67: merge_bn, then quantize_model
        quantize_model(model, json.loads(json.dumps(QCFG)), None, quantization_type="FSPTQ", int8_gemm=args.int8)
    g = torch.Generator(device=dev).manual_seed(2333 + rank)
    x = torch.randn(args.batch, 3, 224, 224, device=dev, generator=g)
    if args.input == "halfnormal":
        x = torch.relu(x)
    # (the image stays NCHW, as a torchvision loader - the reference's, example/classification - hands it over: the first layer's
    #  quantiser reads the three planes with 16-byte loads; a channels_last image costs it 17 us more per step at batch 512)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    args.fused = args.int8 and args.plan == "fused"
    first_batch_ms = None
    from dlmc.quantization.scalar import _wrapper as Wr
    with torch.no_grad():
        model(x)                                 # the first forward calibrates (observer + all-reduce); not timed
        calib_allreduces = Wr.ALLREDUCE_CALLS    # collectives this rank issued in it (one per activation observer when world > 1)
        if args.conv == "int8":
            # SURVEY 8(d) config 3 also asks for the first batch (observers on): re-arm every wrapper's observer and time ONE
            # more calibrating forward (the first one above also paid the one-time costs: code-object loads, allocator growth)
            def rearm():
                for m in model.modules():
                    if hasattr(m, "_init") and hasattr(m, "in_init_state") and hasattr(m, "wt_init_state"):
                        m._init.mark(m, "in_init_state", False)
                        m._init.mark(m, "wt_init_state", False)
            rearm()
            barrier()
            t0 = time.perf_counter()
            model(x)
            barrier()
            first_batch_module_ms = (time.perf_counter() - t0) * 1e3
            # the same calibrating forward with every layer -> (+ shortcut) -> ReLU chain on the int8 route as ONE launch
            # (dlmc.utils.fuse.EagerFused: the wrappers observe and calibrate as above and end with identical scales - tested)
            first_batch_ms, first_batch_how = first_batch_module_ms, "module path"
            from dlmc.utils.fuse import EagerFused
            try:
                eager = EagerFused(model)            # reads the dataflow once (torch.fx), as the plan does
            except RuntimeError as e:                # a model torch.fx cannot trace keeps the module path
                eager = None
                print(f"[bench] EagerFused not used: {e}", file=sys.stderr)
            if eager is not None:                    # (a kernel or launch failure in here is an error of the run, not a fallback)
                rearm()
                eager(x)                             # (one-time costs of this route)
                rearm()
                barrier()
                t0 = time.perf_counter()
                eager(x)
                barrier()
                first_batch_ms, first_batch_how = (time.perf_counter() - t0) * 1e3, "EagerFused"
        if args.fused:
            from dlmc.utils.fuse import fuse_inference
            model = fuse_inference(model, block_layout=not args.row_major_blocks)        # scales are frozen from here on (BASELINE configs[2]: steady state)
        single = model
        if args.fused and args.streams > 1 and not w4a8:     # (QBase plans - grad_scale depends on the elements per call - are not split)
            from dlmc.utils.fuse import StreamedPlan
            model = StreamedPlan(model, args.streams)
        for _ in range(args.warmup):
            model(x)
        if model is not single:
            single(x)
        # Per-kernel HIP events (two marker packets per launch, ~4 us each on the queue) are recorded in the FIRST step of
        # the timed region only: instrumenting all K steps costs the step 5-6 % (9.4 vs 8.85 ms), one step costs 1/K of it.
        K.PROFILE.reset()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            K.PROFILE.enabled = i < args.profiled_steps
            (single if K.PROFILE.enabled else model)(x)
        barrier()
        elapsed = time.perf_counter() - t0
        K.PROFILE.enabled = False

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # per-kernel-family HIP-event durations of the timed region (rank 0's launches)
    fam = families(K.PROFILE.records)
    coll = collective_report(dev, world, args.backend, calib_allreduces) if world > 1 else None

    def other_configs():
        """BASELINE configs[3]'s and [4]'s networks, a short run each after the headline region (every rank takes part)."""
        if not (args.other_configs and args.fused and args.model == "resnet50"):
            return None
        return {name: side_config(name, b, args, dev, world, rank, barrier) for name, b in (("repvgg_a1", 512), ("mobileone_s1", 1024))}
    if rank != 0:
        other_configs()
        if world > 1:
            dist.destroy_process_group()
        return

    images = args.batch * world * args.steps
    psteps = max(1, min(args.profiled_steps, args.steps))   # steps that carried HIP events

    def roof(name, f, kernel, note=None, ops=0):
        ach = f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f["ms"] > 0 else 0.0
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": kernel, "launches": f["launches"],
             "avg_launch_us": round(f["ms"] * 1e3 / max(f["launches"], 1), 2),
             "algorithmic_bytes_per_launch_avg": f["bytes"] // max(f["launches"], 1)}
        if ops and f["ms"] > 0:
            # which roofline binds: the one that needs more time at its peak
            t_hbm, t_mfma = f["bytes"] / (HBM_PEAK_GBPS * 1e9), ops / (MFMA_I8_PEAK_TOPS * 1e12)
            r["mfma"] = {"achieved": round(ops / (f["ms"] * 1e-3) / 1e12, 1), "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                         "frac": round(ops / (f["ms"] * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, 4)}
            r["roofline_ms_per_step"] = {"hbm": round(t_hbm / psteps * 1e3, 3), "mfma": round(t_mfma / psteps * 1e3, 3)}
            if t_mfma > t_hbm:
                r.update(bound="mfma", achieved=r["mfma"]["achieved"], peak=MFMA_I8_PEAK_TOPS, unit="TOP/s", frac=r["mfma"]["frac"])
        if note:
            r["note"] = note
        return r

    empty = {"launches": 0, "bytes": 0, "ms": 0.0, "ops": 0}
    fq = fam.get("fq_tensor", empty)
    fq_roof = roof("fq_tensor", fq, "fq_tensor_kernel<ZEROPOINT> (per-tensor activation fake-quant"
                   + (", int8 code emission: 5 B/elem)" if args.int8 else ": 8 B/elem)"))
    conv = fam.get("conv_i8", empty)              # the generic int8 kernel: 1x1 layers outside chains, dual kernels, fc
    halo = fam.get("conv3x3_halo", empty)         # the halo-tile 3x3 kernel (csrc/conv3x3_i8.hip): its own family since round 4
    chain = fam.get("conv_chain", empty)          # block end + next 1x1 in one kernel (csrc/conv_chain_i8.hip)
    conv_ops, chain_ops = conv.get("ops", 0), chain.get("ops", 0)     # 2 x MACs, recorded per launch (kernels.PROFILE)
    # `roofline` describes the kernel of this project with the largest share of the timed region
    second_roof = third_roof = fourth_roof = None
    if max(conv["ms"], chain["ms"], halo["ms"]) > fq["ms"]:
        if args.fused:
            note = ("algorithmic bytes per launch = int8 input + int8 weights + what the epilogue moves (1 B/elem codes, "
                    "4 B/elem fp32 output where a shortcut / pool needs it, 4 B/elem residual read)")
            conv_roof = roof("conv_i8", conv, "conv_i8_mfma_kernel (int8 implicit-GEMM conv/linear; epilogue: dequant + residual + ReLU + "
                                              "next layer's activation codes)", note, ops=conv_ops)
            chain_roof = roof("conv_chain", chain, "conv_chain_i8_kernel (a block's last 1x1 convolution + shortcut + ReLU + quantiser and the "
                                                   "next block's first 1x1 convolution in one launch; the code tensor between them stays in LDS)",
                              "algorithmic bytes per launch = int8 inputs + int8 weights of both (three) convolutions + 4 B/elem fp32 shortcut read "
                              "+ 4 B/elem fp32 output (1 B/elem codes at stage ends) + the second convolution's codes", ops=chain_ops)
            ranked = [(chain["ms"], chain_roof), (conv["ms"], conv_roof)]
            if halo["ms"] > 0:
                ranked.append((halo["ms"], roof("conv3x3_halo", halo, "conv3x3_halo_i8_kernel (3x3 / pad 1 layers of stride 1 or 2 that emit only their "
                                                "consumer's codes: linear-frame stencil over a halo tile staged once per 64-channel chunk)",
                                                "algorithmic bytes per launch = int8 input + int8 weights + 1 B/elem codes out", ops=halo["ops"])))
            pwr = fam.get("conv_pwr", empty)       # block ends outside the chains on the weight-resident kernel (csrc/conv_pwr_i8.hip)
            if pwr["ms"] > 0:
                ranked.append((pwr["ms"], roof("conv_pwr", pwr, "conv_pwr_i8_kernel (a block's last 1x1 convolution outside the chains: + fp32 shortcut or + a "
                                               "second, sampled 1x1 convolution, ReLU, fp32 output and / or codes; weight slices resident in LDS)",
                                               "algorithmic bytes per launch = int8 input(s) + int8 weights + 4 B/elem fp32 shortcut read + 4 B/elem fp32 output + "
                                               "1 B/elem codes, as the layer has them", ops=pwr["ops"])))
            ranked = [r for ms, r in sorted(ranked, key=lambda t: -t[0]) if ms > 0]
            main_roof, second_roof = ranked[0], (ranked[1] if len(ranked) > 1 else None)
            third_roof = ranked[2] if len(ranked) > 2 else None
            fourth_roof = ranked[3] if len(ranked) > 3 else None
        else:
            main_roof = roof("conv_i8", conv, "conv_i8_mfma_kernel (fused int8-dequant x GEMM conv/linear, fp32 NHWC out)",
                             "algorithmic bytes = int8 input + int8 weights + fp32 output per launch; the 1x1 layers are bound by "
                             "the fp32 output stream, the 3x3 layers by the MFMA pipeline (see conv_i8.TOPs)", ops=conv_ops)
    else:
        main_roof = fq_roof
    dw = fam.get("conv_dw", empty)                 # depthwise 3x3 on codes (csrc/conv_dw_i8.hip): MobileOne
    if args.fused and dw["ms"] > 0:
        dw_roof = roof("conv_dw", dw, "conv_dw3_i8_kernel (depthwise 3x3 on activation codes, fused ReLU + the consumer's codes)",
                       "algorithmic bytes per launch = 1 B/elem codes in + 1 B/elem codes out (+ 4 B/elem where an fp32 output is kept)")
        if dw["ms"] > max(conv["ms"], chain["ms"], halo["ms"]):
            main_roof, second_roof = dw_roof, main_roof
        elif second_roof is None:
            second_roof = dw_roof
    if args.fused and max(conv["ms"], chain["ms"], halo["ms"]) > fq["ms"] and args.model == "resnet50" and args.batch == 512:
        for r in (main_roof, second_roof, third_roof, fourth_roof):
            if r is None:
                continue
            r["traffic"], src, why = pmc_traffic(r["kernel"].split(" ")[0], r["launches"] // psteps, r["algorithmic_bytes_per_launch_avg"])
            r["traffic_measured_in_run"] = False
            if src:
                r["traffic_source"] = f"profiles/{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same command with --no-other-configs)"
                r["traffic_over_algorithmic"] = round(r["traffic"] / max(r["algorithmic_bytes_per_launch_avg"], 1), 4)
            else:
                r["traffic_unavailable"] = why
    if args.fused:
        # In the fused plan the activation quantiser lives in the conv epilogue, so the stand-alone fake-quant kernel
        # hardly appears in the step.  Its own roofline is measured here, live, on BASELINE configs[1]'s tensor
        # (A = 64 x 256 x 56 x 56 fp32, per-channel axis 1: 8 algorithmic bytes per element), HIP events per launch.
        from dlmc import _native as Nn
        a = torch.randn(64, 256, 56, 56, device=dev)
        sc = a.abs().amax(dim=(0, 2, 3)) / 127 + 1e-6
        zs = torch.zeros_like(sc)
        for _ in range(3):
            K.fake_quant(a, sc, zs, -127, 127, Nn.FORM_EMULATE, ch_axis=1)
        K.PROFILE.enabled = True
        K.PROFILE.reset()
        for _ in range(20):
            K.fake_quant(a, sc, zs, -127, 127, Nn.FORM_EMULATE, ch_axis=1)
        torch.cuda.synchronize()
        K.PROFILE.enabled = False
        f2 = {"launches": 0, "bytes": 0, "ms": 0.0}
        for tag, nbytes, e0, e1, _ in K.PROFILE.records:
            f2["launches"] += 1
            f2["bytes"] += nbytes
            f2["ms"] += e0.elapsed_time(e1)
        fq_roof = roof("fq_channel", f2, "fq_channel_kernel<EMULATE> stand-alone on BASELINE configs[1] (64x256x56x56 fp32, per-channel "
                                         "axis 1, 4 B read + 4 B written per element); in the fused plan this arithmetic runs in the conv epilogue")
        del a
    qbytes = sum(f["bytes"] for k, f in fam.items() if k.startswith("fq"))
    qms = sum(f["ms"] for k, f in fam.items() if k.startswith("fq"))
    out = {
        "metric": "ResNet-50 W8A8 fake-quant fwd images/sec" if args.model == "resnet50" else
                  f"{args.model} {'W4A8' if w4a8 else 'W8A8'} fake-quant fwd images/sec",
        "value": round(images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None,
        # the arithmetic of the step: int8 x int8 -> int32 on the matrix cores with fp32 quantise / dequantise (int8 mode), or fp32
        "dtype": "i8" if args.int8 else "f32", "data": "synthetic",
        "config": {"workload": (f"{args.model} W4A8 fake-quant forward (QBase family: W minmax_channel u4 asymmetric per channel, "
                                f"A minmax_tensor u8), deploy form, " if w4a8 else
                                f"{args.model} W8A8 per-channel fake-quant forward (FSPTQ forms: W minmax_channel s8, "
                                f"A minmax_tensor u8), {'BatchNorm kept' if args.keep_bn else 'BN folded first as in FSPTQuant.py:67'}, ") +
                               f"{'fused int8 MFMA conv/linear' if args.int8 else 'fp32 conv of the fake-quantised operands'}, "
                               f"{'frozen execution plan (epilogue-fused ReLU / shortcut / next-layer codes; block end + next 1x1 as one launch; weight codes quantised once at plan build), ' if args.fused else ''}"
                               f"{str(args.streams) + ' HIP streams per GPU (the ' + str(min(args.profiled_steps, args.steps)) + ' profiled step(s) on one), ' if args.fused and args.streams > 1 and not w4a8 else ''}"
                               f"224x224 {'relu(N(0,1))' if args.input == 'halfnormal' else 'N(0,1)'} pixels, batch {args.batch} per GPU, scales frozen",
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (batch-sharded replicas)"},
        **({"first_batch": {"ms": round(first_batch_ms, 2), "images_per_s": round(args.batch * world / (first_batch_ms * 1e-3), 1),
                            "how": first_batch_how, "module_path_ms": round(first_batch_module_ms, 2),
                            "what": "one forward with every observer on (min/max and all-reduce(MAX) per activation quantiser, per-channel "
                                    "weight scales, then the layer); `how` = EagerFused: layer + shortcut add + ReLU are one int8 launch wherever "
                                    "the layer is on the int8 route, the consumer's min/max comes out of that launch's epilogue where the tiled "
                                    "kernel ran (round 5: no second read of the tensor), and the zero points are checked with ONE host read per "
                                    "forward (same scales and outputs as the module path, whose time is `module_path_ms`: torch's ReLU / add "
                                    "between the layers, a min/max pass and a host read per layer); host-launch-bound; not part of `value`"}}
           if first_batch_ms is not None else {}),
        "roofline": main_roof,
        **({"roofline_second_kernel": second_roof} if second_roof else {}),
        **({"roofline_third_kernel": third_roof} if third_roof else {}),
        **({"roofline_fourth_kernel": fourth_roof} if fourth_roof else {}),
        "roofline_fake_quant": fq_roof,
        "quant_path": {"images_per_s": round(args.batch * psteps / (qms * 1e-3), 1) if qms else None,
                       "GBps": round(qbytes / (qms * 1e-3) / 1e9, 1) if qms else None,
                       "ms_per_step": round(qms / psteps, 3),
                       "share_of_step": round(qms / psteps / (elapsed / args.steps * 1e3), 4),
                       "families": {k: {"launches": f["launches"], "GBps": round(f["bytes"] / (f["ms"] * 1e-3) / 1e9, 1)}
                                    for k, f in fam.items() if f["ms"] > 0}},
    }
    if args.model == "resnet50":
        # 85 315 584 B per image of activations + 204 023 296 B of weights per forward at 8 B per
        # fake-quantised element; "5.49 ms at roofline" for batch 512.  The whole step (convolutions included) against that.
        qb = 85315584 * args.batch + 204023296
        step_s = elapsed / args.steps
        out["quant_work_8d"] = {"algorithmic_bytes_per_step_per_gpu": qb, "equivalent_GBps": round(qb / step_s / 1e9, 1),
                                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": round(qb / step_s / 1e9 / HBM_PEAK_GBPS, 4),
                                "note": "SURVEY.md 8(d) accounting: bytes the reference's stand-alone fake-quant passes move per step "
                                        "/ the whole step's time (convolutions included); an equivalent rate - in the fused plan "
                                        "the quantisers run in the conv epilogues and most of these bytes never exist"}
    if conv["ms"] + chain["ms"] + halo["ms"] > 0:     # the int8 matrix-core kernels together
        ms3 = conv["ms"] + chain["ms"] + halo["ms"]
        out["conv_i8"] = {"launches": conv["launches"] + chain["launches"] + halo["launches"], "ms_per_step": round(ms3 / psteps, 3),
                          "GBps": round((conv["bytes"] + chain["bytes"] + halo["bytes"]) / (ms3 * 1e-3) / 1e9, 1),
                          "TOPs": round((conv_ops + chain_ops + halo.get("ops", 0)) / (ms3 * 1e-3) / 1e12, 1),
                          "peak_TOPs_dense_i8": MFMA_I8_PEAK_TOPS}
    if coll is not None:
        out["collective"] = coll
    if args.fq_leg and args.fused and args.model == "resnet50" and world == 1:
        out["roofline_fake_quant_resnet50"] = fake_quant_leg(args, dev, x)
    other = other_configs()
    if other is not None:
        out["other_configs"] = other
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_batch, budget_s=args.cpu_budget, fold_bn=not args.keep_bn,
                                           halfnormal=args.input == "halfnormal")
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
