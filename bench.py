#!/usr/bin/env python
"""bench.py -- policy steps/s of the MI355X-native FastVLA path (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: bench.py starts its own N ranks, one process per GPU, before it
                                           touches the GPU itself, and relays rank 0's JSON line)

One "step" = one batch through img + prompt + state -> action (BASELINE.json configs[1]: FastVLM-0.5B select_action,
bs=64 per GPU, 336x336 synthetic RGB + 64-token prompt, bf16 MFMA with fp32 accumulation): letterbox to 1024^2 ->
FastViT-HD -> mm_projector -> Qwen2 decoder -> last-token pool -> action expert, inputs resident in HBM.
Inference has no exchange step, so N ranks are N replicas on disjoint batches (weak scaling, no collective); the
data-parallel TRAINING step (B=32/GPU, RCCL all-reduce of the flat head gradient) is timed as well and reported under
"train_dp" in the same JSON line.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (the fused ConvFFN, convffn32_kernel): algorithmic FLOPs / HIP-event time per launch, against
                the 2.5 PFLOP/s dense bf16 peak
  cpu_baseline  the fp32 CPU oracle (oracle/, "port") timed on this box's host cores on a bounded sample
  surface       the same step through the reference's plugin surface (lerobot_fastvla.FastVLAPolicy.select_action / forward,
                task strings through the tokenizer), beside the engine-level number
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "vla-from-fastvlm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0



def _usable_cpus() -> int:
    """Threads of the CPU baseline legs: EVERY host CPU this process may run on (set_num_threads(os.cpu_count())) -- the scheduler
    affinity set, cut to the cgroup's CPU quota where the box grants a share of its cores (more runnable threads than granted CPUs only makes the baseline
    slower).  FASTVLA_CPU_THREADS overrides it.  The count used is printed in the line ("cores")."""
    if os.environ.get("FASTVLA_CPU_THREADS"):
        return max(1, int(os.environ["FASTVLA_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="fastvlm-0.5b")
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch of the inference step")
    ap.add_argument("--train-batch", type=int, default=32, help="per-GPU batch of the DP training step")
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--image", type=int, default=336)
    ap.add_argument("--splice", action="store_true", help="splice the 256 projected image tokens into the LLM sequence")
    ap.add_argument("--microbatch", type=int, default=int(os.environ.get("FASTVLA_TOWER_MICROBATCH", "0")))
    ap.add_argument("--llm-precision", type=int, default=None, choices=(0, 1, 2, 5),
                    help="1 (default for every model: arch.default_llm_precision) = split-bf16 decoder operands + fp32 attention (actions ~1e-5 from the "
                         "fp32 reference, every row inside north_star's 1e-3); 2 = opt-in: split-bf16 qkv / o + ONE fp16 pass for gate/up and down "
                         "(actions 4.4e-4 .. 6.4e-4 as a batch rel-L2, but the worst ROW of C1 measures 1.1e-3: tests/test_gpu_fullsize.py); "
                         "0 = plain bf16 operands (~8e-3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-train-unfrozen", action="store_true", help="skip the unfrozen decoder + projector training leg (SURVEY.md 8f-4)")
    ap.add_argument("--no-train-tower", action="store_true", help="skip the everything-trainable leg (tower backward too)")
    ap.add_argument("--train-unfrozen-dp", action="store_true", help="(accepted for compatibility: at N > 1 the unfrozen training legs run by default since round 6 -- the ranks agree on "
                                                                    "their set-up before the first collective and --secondary-deadline bounds the whole tail; --no-train-unfrozen-dp skips them)")
    ap.add_argument("--unfrozen-batch", type=int, default=int(os.environ.get("FASTVLA_UNFROZEN_BATCH", "32")), help="per-GPU batch of the unfrozen training leg (C3's rank shape)")
    ap.add_argument("--fv-comm-check", action="store_true", help="(accepted for compatibility: the check below is on by default since round 6)")
    ap.add_argument("--no-fv-comm-check", action="store_true",
                    help="N > 1 on the RCCL backend: skip the live check of the library's own communicator (fv_comm_* + one all-reduce of ones through "
                         "fv_allreduce_grads).  By default it runs LAST, after every measurement, on a helper thread with a time limit: a failure or a stall is "
                         "recorded in dist.fv_comm and can neither abort nor hang the line")
    ap.add_argument("--no-train-unfrozen-dp", action="store_true", help="N > 1: skip the unfrozen training legs (per-bucket all-reduce under the backward pass)")
    ap.add_argument("--secondary-deadline", type=float, default=float(os.environ.get("FASTVLA_BENCH_DEADLINE", "420")),
                    help="N > 1: seconds the legs after the headline measurement may take in total; past it rank 0 prints the line with what it has "
                         "(dist.deadline_hit = true) and every rank exits -- a secondary leg can delay a scaling run, not lose it")
    ap.add_argument("--no-alt", action="store_true", help="skip the second engine that times the OTHER decoder parity mode (llm_precision 1 <-> 2)")
    ap.add_argument("--no-surface", action="store_true", help="skip the plugin-surface leg (FastVLAPolicy.select_action / forward)")
    ap.add_argument("--cpu-sample", type=int, default=6)   # ~12 s of host work on 16 threads
    ap.add_argument("--no-c1", action="store_true", help="skip the C1 (bs=4, 32-token, train step) CPU protocol of BASELINE.md section 3")
    ap.add_argument("--c1-steps", type=int, default=10, help="C1 protocol: steps requested (first excluded)")
    ap.add_argument("--c1-budget", type=float, default=60.0, help="C1 protocol: stop starting new CPU steps after this many seconds")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only to "
                                                      "rehearse the multi-process path on a one-GPU box)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher (the form the reference gets from `accelerate launch`,
    training/trainer.py:68-78): start N fresh rank processes -- one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment -- wait for them, relay rank 0's JSON line.  Runs BEFORE this process makes any HIP call or torch.cuda query and
    never exec()s: the parent stays a plain host process."""
    import socket
    import subprocess
    import threading
    import time
    # the rendezvous port: kept bound (SO_REUSEADDR) until just before the ranks start, so nothing else grabs it in between
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # drain rank 0 while polling
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("FASTVLA_BENCH_TIMEOUT_S", "1500"))
    codes = [None] * len(procs)
    failed = False
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        bad_now = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad_now or time.monotonic() > deadline:
            # one rank died (OOM, bad device) or the job hangs: the others would sit in init_process_group / a barrier until the
            # store or RCCL timeout -- end exactly the children this function started and report a fast non-zero exit
            failed = True
            t_grace = time.monotonic() + 3.0      # ranks that fail for the same reason report their OWN exit code
            while time.monotonic() < t_grace and any(c is None for c in codes):
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        codes[r] = p.poll()
                time.sleep(0.1)
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            t_kill = time.monotonic() + 10
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            why = f"rank(s) {bad_now} exited non-zero" if bad_now else "overall timeout"
            print(f"bench.py: {why}; remaining ranks terminated", file=sys.stderr)
            break
        time.sleep(0.2)
    reader.join(timeout=5)
    sys.stdout.write("".join(x for x in out0 if x))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or failed:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)  # > 1 rank per GPU only in gloo rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)

    from fastvla_hip import FastVLAEngine, arch, weights
    model = arch.preset(args.model)
    if args.llm_precision is None:
        args.llm_precision = arch.default_llm_precision(model)
    B, T = args.batch, args.tokens
    eng = FastVLAEngine(model, state_dim=14, action_dim=14, hidden_dim=1024, fusion_dim=1024, device=dev,
                        max_batch=max(B, args.train_batch), max_text_tokens=T, tower_microbatch=args.microbatch,
                        llm_precision=args.llm_precision)
    t0 = time.time()
    big = 3 * model.llm.hidden * model.llm.inter * model.llm.layers > 2e9   # 7B: 7.6 G parameters = 30 GB as an fp32 host dict
    if big or world > 1:
        # streamed: every decoder tensor is drawn on the device in bf16 when the packer asks for it (fv_load_weights_cb);
        # identical on every rank (seeded by name).  No host copy exists, so the CPU-oracle legs are skipped for this model --
        # and N ranks of one host do not each build a 2.5 GB fp32 dict with all host cores at once (the oracle legs are N = 1 only)
        w = None
        eng.load_weights_streaming(weights.stream_backbone(model, seed=args.seed, device=dev))
    else:
        w = weights.init_backbone(model, seed=args.seed)  # identical on every rank (frozen replica)
        eng.load_weights(w)
    t_load = time.time() - t0

    # what the communication layer itself reports (N > 1): torch.distributed's world and, with --fv-comm-check on the RCCL backend, a
    # communicator made through the library's own C ABI (fv_comm_*) with one all-reduce through fv_allreduce_grads as a live check
    dist_info = None
    if world > 1:
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank0_device": torch.cuda.get_device_name(dev), "deadline_hit": False,
                     "fv_comm": None}   # filled by the live check at the very end of the run (see fv_comm_live_check)

    # trainable head: torch.nn default-style init, identical on every rank
    g = torch.Generator().manual_seed(4321)
    flat = torch.zeros(eng.head_numel(), dtype=torch.float32, device=dev)
    for k, v in eng.head_views(flat).items():
        if v.ndim == 2:
            bound = 1.0 / (v.shape[1] ** 0.5)
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * bound)
        elif k in ("state_projection.0.weight", "fusion.1.weight"):
            v.fill_(1.0)
        elif k.endswith("bias") and k not in ("state_projection.0.bias", "fusion.1.bias"):
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * 0.02)

    torch.manual_seed(1234 + rank)
    Bmax = max(B, args.train_batch)
    images = torch.rand(Bmax, 3, args.image, args.image, device=dev)
    ids = torch.randint(0, min(151643, model.llm.vocab), (Bmax, T), device=dev, dtype=torch.int32)
    lens = torch.full((Bmax,), T, dtype=torch.int32, device=dev)
    states = torch.randn(Bmax, 14, device=dev)
    targets = torch.randn(Bmax, 14, device=dev)

    def step_infer():
        pooled = eng.backbone(images[:B], ids[:B], lens[:B], splice=args.splice)
        act, _ = eng.head_forward(flat, pooled, states[:B])
        return act

    def barrier():
        if world > 1:
            dist.barrier()

    # board power of THIS rank's GPU while the timed region runs (amdgpu hwmon, microwatts; no subprocess): whether the step sits at
    # the board's cap is what decides if a kernel-level cycle saving can show up as time at all
    class PowerSampler:
        def __init__(self):
            self.path = self.cap = self.last = None
            self.samples = []
            self._stop = threading.Event()
            self._th = None
            try:
                pr = torch.cuda.get_device_properties(dev)
                slot = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
                for ue in sorted(glob.glob("/sys/class/drm/card*/device/uevent")):
                    if f"PCI_SLOT_NAME={slot}" in open(ue).read():
                        hw = glob.glob(os.path.join(os.path.dirname(ue), "hwmon", "hwmon*", "power1_input"))
                        if hw:
                            self.path = hw[0]
                            self.cap = int(open(hw[0].replace("power1_input", "power1_cap")).read()) / 1e6
                        break
            except Exception:
                self.path = None

        def _run(self):
            while not self._stop.is_set():
                try:
                    self.samples.append(int(open(self.path).read()) / 1e6)
                except Exception:
                    pass
                self._stop.wait(0.02)

        def start(self):
            if self.path:
                self.samples, self._th = [], threading.Thread(target=self._run, daemon=True)
                self._stop.clear()
                self._th.start()

        def stop(self):
            if self._th:
                self._stop.set()
                self._th.join()
                self._th = None
            if not self.samples:
                return None
            sm = sorted(self.samples)
            return {"board_w_mean": round(sum(sm) / len(sm), 1), "board_w_median": round(sm[len(sm) // 2], 1), "board_w_max": round(sm[-1], 1),
                    "cap_w": self.cap, "samples": len(sm), "source": "amdgpu hwmon power1_input of rank 0's GPU, sampled every 20 ms inside the timed region"}

    power = PowerSampler() if rank == 0 else None

    def timed(fn, steps, warmup, sampler=None):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        if sampler:
            sampler.start()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t
        if sampler:
            sampler.last = sampler.stop()
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt)
        return el

    el = timed(step_infer, args.steps, args.warmup, power)
    power_info = power.last if power else None
    ms_per_step = 1e3 * el / args.steps
    value = world * args.steps / el

    # ---- per-kernel-family HIP-event pass (same work, right after the timed region, on the same stream)
    # kernels are timed one at a time here: the decoder/tower stream overlap of the timed region would make concurrent
    # kernels share the chip and inflate each other's durations (the committed rocprofv3 stats use FASTVLA_OVERLAP=0 too)
    overlap = eng.overlap_streams
    eng.overlap_streams = False
    eng.profile(True)
    for _ in range(args.profile_steps):
        step_infer()
    torch.cuda.synchronize()
    fams, shapes = eng.profile_read()
    eng.profile(False)
    eng.overlap_streams = overlap
    ps = args.profile_steps
    gem = fams["gemm"]
    gemm_tflops = gem["flops"] / (gem["ms"] * 1e-3) / 1e12 if gem["ms"] > 0 else 0.0
    total_flops = sum(f["flops"] for f in fams.values()) / ps
    # the MFMA family is two kernels: the fused ConvFFN (epi 6: 4*M*C*4C flop) and the plain GEMM (2*M*N*|K|)
    def _fl(r):
        return (4.0 if r["epi"] == 6 else 2.0) * r["m"] * r["n"] * abs(r["k"]) * r["launches"]
    ffn_name = "convffn32_kernel (fused fc1+GELU+fc2, bf16 MFMA 32x32x16)"
    # `roofline` is ONE kernel as rocprofv3 lists it: the fused ConvFFN is a template with one instance per channel width, each its own
    # row of the --stats table, so the dominant kernel is the instance with the most time (C = 384 at the 0.5B tower), not the three
    # pooled; the pooled, launch-weighted figure of rounds 1-2 stays beside it as "family"
    ffn_rows = [r for r in shapes if r["epi"] == 6]
    groups = {f"{ffn_name.split(' ')[0]}<C={c}> {ffn_name.split(' ', 1)[1]}": [r for r in ffn_rows if r["n"] == c] for c in sorted({r["n"] for r in ffn_rows})}
    groups["gemm_kernel (bf16 MFMA 16x16x32, fp32 acc)"] = [r for r in shapes if r["epi"] != 6]
    gstat = {k: dict(ms=sum(r["ms"] for r in v), flops=sum(_fl(r) for r in v), n=sum(r["launches"] for r in v))
             for k, v in groups.items() if v}
    dom = max((k for k in gstat if not k.startswith("gemm_kernel")), key=lambda k: gstat[k]["ms"], default=None)
    if dom is None or gstat["gemm_kernel (bf16 MFMA 16x16x32, fp32 acc)"]["ms"] > sum(v["ms"] for k, v in gstat.items() if not k.startswith("gemm_kernel")):
        dom = "gemm_kernel (bf16 MFMA 16x16x32, fp32 acc)"
    dg = gstat[dom]
    dom_tflops = dg["flops"] / (dg["ms"] * 1e-3) / 1e12
    fam = dict(ms=sum(r["ms"] for r in ffn_rows), flops=sum(_fl(r) for r in ffn_rows), n=sum(r["launches"] for r in ffn_rows)) if ffn_rows else None
    traffic = None
    traffic_src = None
    pmc = ROOT / "profiles" / "pmc_traffic.json"  # HBM bytes per launch from rocprofv3 --pmc passes (collected offline)
    if pmc.is_file():
        try:
            traffic = json.loads(pmc.read_text()).get(dom.split(" ")[0].replace("<C=", "<"))   # per template instance (tools/pmc_traffic.py)
            traffic_src = "profiles/pmc_traffic.json: rocprofv3 --pmc passes of an EARLIER run of this command (tools/pmc_traffic.py), not a quantity of this run"
        except Exception:
            traffic = None
    roofline = {
        "bound": "mfma", "kernel": dom,
        "achieved": round(dom_tflops, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(dom_tflops / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
        # both readings at the TOP level so rounds stay comparable: `frac_instance` = this line's `frac` (the dominant template instance as
        # rocprofv3 lists it; rounds 3+), `frac_family` = the fused ConvFFN over all channel widths, launch-weighted (what rounds 1-2 called frac)
        "frac_instance": round(dom_tflops / MFMA_PEAK_TFLOPS, 4),
        "frac_family": None if not fam else round(fam["flops"] / (fam["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
        "practical": {"peak": 1860.0 if "32x32x16" in dom else 2050.0, "frac": round(dom_tflops / (1860.0 if "32x32x16" in dom else 2050.0), 4), "unit": "TFLOP/s",
                      "source": "tools/mfma_ceiling.hip measured on this chip in round 2: the bare MFMA loop of this kernel's shape on random bf16 operands held in registers, "
                                "i.e. what the board's 1.4 kW cap buys with nothing else switching (DESIGN.md 5.0, 5.5); a constant of the repo, not a quantity of this run"},
        "launches_per_step": dg["n"] // ps, "kernel_ms_per_step": round(dg["ms"] / ps, 3),
        "kernel_flops_per_launch": dg["flops"] / dg["n"], "kernel_avg_launch_ms": round(dg["ms"] / dg["n"], 4),
        "family": None if not fam else {"kernels": ffn_name + ", all channel widths, launch-weighted (the figure rounds 1-2 reported as roofline)",
                                        "achieved": round(fam["flops"] / (fam["ms"] * 1e-3) / 1e12, 2),
                                        "frac": round(fam["flops"] / (fam["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                        "launches_per_step": fam["n"] // ps, "ms_per_step": round(fam["ms"] / ps, 3),
                                        "avg_launch_ms": round(fam["ms"] / fam["n"], 4),
                                        "per_width": {k.split(" ")[0]: {"frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                                                        "ms_per_step": round(v["ms"] / ps, 3), "avg_launch_ms": round(v["ms"] / v["n"], 4)}
                                                      for k, v in gstat.items() if not k.startswith("gemm_kernel")}},
        "all_mfma_kernels": {"achieved": round(gemm_tflops, 2), "frac": round(gemm_tflops / MFMA_PEAK_TFLOPS, 4),
                             "ms_per_step": round(gem["ms"] / ps, 3), "flops_per_step": gem["flops"] / ps},
        "step_achieved": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
        "step_frac": round(total_flops / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
        "source": f"hipEvents around every launch over {ps} steps after the timed region (stream overlap off); algorithmic flops "
                  "(2MNK per GEMM, 4*M*C*4C per fused ConvFFN; split-bf16 passes not double-counted)",
    }
    families = {k: {"ms_per_step": round(v["ms"] / ps, 3), "launches": v["launches"] // ps,
                    "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2),
                    "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in fams.items() if v["launches"]}
    top = sorted(shapes, key=lambda r: -r["ms"])[:30]
    gemm_shapes = [{"mnk": [r["m"], r["n"], r["k"]], "epi": r["epi"], "ms_per_step": round(r["ms"] / ps, 3),
                    "n": r["launches"] // ps,
                    "tflops": round(_fl(r) / max(r["ms"], 1e-9) / 1e9, 1)} for r in top]

    # ---- the line, as a closure over everything measured so far: the legs below fill their own fields in.  At N > 1 a watchdog bounds them -- whatever a
    # secondary leg does on hardware nobody has run it on yet (a stalled collective, one rank failing alone), the headline of a scaling run still gets printed.
    power_info_ = power_info
    train = train_unfrozen = train_unfrozen_tower = surface = prefix = alt = cpu = c1 = None
    emitted = threading.Lock()

    def emit():
        if rank != 0 or not emitted.acquire(blocking=False):
            return
        out = {
            "metric": "policy steps/sec (img+prompt->action) FastVLM-0.5B bs=64" if args.model == "fastvlm-0.5b" and B == 64
                      else f"policy steps/sec (img+prompt->action) {args.model} bs={B}",
            "value": round(value, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (seeded random weights + inputs; no checkpoint/dataset reachable offline)",
            "config": {"workload": f"{args.model} select_action: letterbox {args.image}^2->{model.tower.image_size}^2, FastViT-HD, "
                                   f"projector, Qwen2 decoder ({'256 image tokens + ' if args.splice else 'text-only, reference-literal: '}"
                                   f"{T}-token prompt), last-token pool, action head",
                       "batch_per_gpu": B, "global_batch": B * world, "seq_len": T, "image": args.image,
                       "parallelism": f"replicas x{world} (no collective on the inference path)",
                       "splice_image_tokens": bool(args.splice), "tower_microbatch": args.microbatch,
                       "stream_overlap": bool(eng.overlap_streams and not args.splice),
                       "llm_precision": {0: "bf16 operands", 1: "split-bf16 (hi+lo) operands, fp32 attention",
                                         2: "split-bf16 qkv/o, fp16 gate/up/down (one pass), fp32 attention",
                                         3: "split-bf16 qkv/o/down, fp16 gate/up (one pass), fp32 attention",
                                         4: "split-bf16 qkv/o/gate/up, fp16 down (one pass), fp32 attention",
                                         5: "bf16 hi + fp8 lo operands (lo product on the scaled fp8 MFMA, 1.5 passes), fp32 attention"}[args.llm_precision]},
            "samples_per_s": round(value * B, 2),
            "roofline": roofline, "power": power_info_, "cpu_baseline": cpu, "cpu_baseline_c1": c1, "train_dp": train, "train_unfrozen": train_unfrozen, "train_unfrozen_tower": train_unfrozen_tower, "surface": surface, "dist": dist_info, "splice_prefix_cache": prefix, "other_parity_mode": alt,
            "families": families, "gemm_shapes": gemm_shapes, "weights_load_s": round(t_load, 1),
        }
        print(json.dumps(out), flush=True)

    watchdog = None
    if world > 1:
        def on_deadline():
            dist_info["deadline_hit"] = True
            dist_info["deadline_s"] = args.secondary_deadline
            emit()
            sys.stdout.flush()
            os._exit(0)     # a rank stuck in a collective cannot be joined: every rank runs this timer and leaves by itself
        watchdog = threading.Timer(args.secondary_deadline, on_deadline)
        watchdog.daemon = True
        watchdog.start()

    def ranks_agree(ok: bool) -> bool:
        """N > 1: a leg starts its collectives only if EVERY rank finished the leg's set-up (allocations are where one rank fails alone)."""
        if world == 1:
            return ok
        t_ = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(t_, op=dist.ReduceOp.MIN)
        return bool(int(t_) == 1)

    # ---- data-parallel training step (C3): forward + MSE + head backward + all-reduce + clip + AdamW.
    # Pipelined as vla_fastvlm.training.Trainer runs it: the frozen backbone forward of batch k+1 is enqueued between the
    # START of batch k's gradient all-reduce (side stream, RCCL over xGMI) and the optimiser kernel that needs its result,
    # so the collective runs underneath ~30 ms of tower kernels.  One timed step = one backbone forward + one head step.
    if not args.no_train:
        from vla_fastvlm.training.dp import GradExchange
        Bt = args.train_batch
        m_buf, v_buf = torch.zeros_like(flat), torch.zeros_like(flat)
        grads = torch.zeros_like(flat)
        saved = eng.head_saved(Bt)
        ex = GradExchange(dev)
        state = {"step": 0, "pooled": None, "timed": False, "ar": []}

        def frozen_forward():
            return eng.backbone(images[:Bt], ids[:Bt], lens[:Bt], splice=args.splice)

        def head_step(pooled):
            state["step"] += 1
            act, _ = eng.head_forward(flat, pooled, states[:Bt], training=True, dropout_p=0.1, seed=args.seed + rank,
                                      offset=state["step"], saved=saved)
            eng.head_backward(flat, act, targets[:Bt], saved, dropout_p=0.1, flat_grads=grads)

        def optimiser(scale):
            eng.adamw_step(flat, grads, m_buf, v_buf, state["step"], lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0, grad_scale=scale)

        def step_train():           # pipelined: exchange(k) || frozen forward(k+1)
            head_step(state["pooled"])
            scale = ex.start(grads, timed=state["timed"])
            state["pooled"] = frozen_forward()
            ex.finish(dev)
            if state["timed"] and world > 1:
                state["ar"].append(ex.last_ms())
            optimiser(scale)

        def step_train_serial():    # the unpipelined order, for the overlap figure
            head_step(frozen_forward())
            scale = ex.start(grads)
            ex.finish(dev)
            optimiser(scale)

        flat_backup = flat.clone()
        nst = max(2, args.steps // 2)
        state["pooled"] = frozen_forward()
        elt = timed(step_train, nst, max(1, args.warmup // 2))
        ar_ms = overlap = ser_ms = None
        if world > 1:
            els = timed(step_train_serial, nst, 2)
            ser_ms = 1e3 * els / nst
            state["timed"] = True
            state["pooled"] = frozen_forward()
            for _ in range(3):
                step_train()
            torch.cuda.synchronize()
            state["timed"] = False
            ar_ms = sum(state["ar"]) / max(1, len(state["ar"]))
            tt = torch.tensor([ar_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ar_ms = float(tt)
            # share of the collective's duration that no longer extends the step: (serial - pipelined) / all-reduce time
            overlap = max(0.0, min(1.0, (ser_ms - 1e3 * elt / nst) / ar_ms)) if ar_ms > 0 else None
        flat.copy_(flat_backup)
        train = {"value": round(world * nst / elt, 3), "unit": "train steps/s (whole job)", "ms_per_step": round(1e3 * elt / nst, 3),
                 "global_batch": Bt * world, "parallelism": f"dp{world}",
                 "collective": f"{args.backend} all_reduce(flat head grads), started before the next batch's frozen forward" if world > 1 else None,
                 "grad_bytes": int(flat.numel() * 4), "allreduce_ms": None if ar_ms is None else round(ar_ms, 4),
                 "ms_per_step_serial_exchange": None if ser_ms is None else round(ser_ms, 3),
                 "overlap_frac": None if overlap is None else round(overlap, 3)}

    # ---- unfrozen decoder + projector training (fv_train_* -- the reference's freeze_backbone=False, which its own
    # no_grad forward makes moot): one step = letterbox + FROZEN tower + projector / spliced 320-token decoder forward with every activation
    # kept + MSE + backward over 494 M parameters (dgrad / wgrad on the forward's GEMM kernels, split-bf16 gradient operands) + clip + AdamW +
    # bf16 operand refresh.  Under N > 1 the gradient travels per bucket (one decoder layer = 60 MB) on a side stream while the backward
    # pass is still running.  Its own roofline fraction: algorithmic flops (tower + 3 x (projector + decoder GEMMs) + attention fwd/bwd).
    if not args.no_train_unfrozen and (world == 1 or not args.no_train_unfrozen_dp) and args.llm_precision == 1 and model.llm.head_dim >= 64 and not big:
        from vla_fastvlm.training.dp import BucketedGradExchange
        Bu = min(args.unfrozen_batch, B)
        try:
            setup_err = None
            try:
                eng.train_begin()
                tensors_u, total_u, nb_u = eng.train_layout()
                flat_u = torch.zeros(total_u, dtype=torch.float32, device=dev)
                eng.train_export_params(flat_u)
                flat_u[: flat.numel()].copy_(flat)
                flat_u0 = flat_u.clone()              # the weights as loaded: committed back after the leg (bf16 -> fp32 -> bf16 is exact)
                g_u, m_u, v_u = torch.zeros_like(flat_u), torch.zeros_like(flat_u), torch.zeros_like(flat_u)
                ws_u = eng.train_workspace(Bu, T)
                bex = BucketedGradExchange(dev, min_numel=1 << 22)
            except Exception as exc_:
                setup_err = exc_
            if not ranks_agree(setup_err is None):
                raise RuntimeError(f"set-up failed on {'this' if setup_err is not None else 'another'} rank: {setup_err!r}")
            ust = {"step": 0}

            def step_unfrozen():
                ust["step"] += 1
                pix_u = eng.preprocess(images[:Bu])
                _, tower_out_u = eng.vision_forward(pix_u, return_tower_out=True)
                bex.begin(g_u)
                eng.train_forward_backward(flat_u, tower_out_u, ids[:Bu], lens[:Bu], states[:Bu], targets[:Bu], ws_u, training=True, dropout_p=0.1,
                                           seed=args.seed + rank, offset=ust["step"], flat_grads=g_u, bucket_cb=bex.bucket_ready)
                scale = bex.finish(dev)
                eng.adamw_step(flat_u, g_u, m_u, v_u, ust["step"], lr=1e-5, weight_decay=1e-4, max_grad_norm=1.0, grad_scale=scale / eng.train_loss_scale())
                eng.train_commit(flat_u)

            nsu = max(2, args.steps // 4)
            elu = timed(step_unfrozen, nsu, 2)
            exch_u = None
            if world > 1:    # the same step with the collectives left out: what the per-bucket exchange still EXPOSES (0 = fully hidden under the backward pass)
                bex.dry_run = True
                elu_dry = timed(step_unfrozen, nsu, 1)
                bex.dry_run = False
                exch_u = {"ms_per_step_without_exchange": round(1e3 * elu_dry / nsu, 3), "exchange_exposed_ms": round(1e3 * (elu - elu_dry) / nsu, 3)}
            eng.train_set_forward_f16(True)                           # opt-in: the training forward's projections in ONE fp16 pass (half the forward's MFMA work)
            elu_f16 = timed(step_unfrozen, nsu, 1)
            eng.train_set_forward_f16(False)
            eng.train_set_options(grad_split=1)                       # split-bf16 dgrad operands (two passes): the most exact form
            elu_bf = timed(step_unfrozen, nsu, 1)
            eng.train_set_options(grad_split=1, wgrad_f16=False)      # round 4's first form: + weight gradients as split-bf16 gradient x bf16 activation (two passes)
            elu_w2 = timed(step_unfrozen, nsu, 1)
            eng.train_set_options()
            Ni_u = model.tower.num_tokens
            rows_u = Bu * (Ni_u + T)
            ll = model.llm
            qkvw_u = (ll.heads + 2 * ll.kv_heads) * ll.head_dim
            dec_tok = 2.0 * ll.layers * (ll.hidden * qkvw_u + ll.heads * ll.head_dim * ll.hidden + 3 * ll.hidden * ll.inter)   # GEMM flops per token, lm_head excluded
            dec_fl = dec_tok * rows_u
            proj_fl = 2.0 * Bu * Ni_u * (model.tower.out_dim * ll.hidden + ll.hidden ** 2)
            attn_fl = 2.0 * 2 * Bu * (Ni_u + T) ** 2 / 2 * ll.heads * ll.head_dim * ll.layers   # QK^T + PV, causal half
            # the tower's algorithmic flops per image from the inference step's own per-launch accounting (literal mode: tower + projector + T-token decoder)
            tower_fl = max(0.0, total_flops - dec_tok * B * T - 2.0 * B * Ni_u * (model.tower.out_dim * ll.hidden + ll.hidden ** 2)) / B * Bu
            step_fl = tower_fl + 3.0 * (dec_fl + proj_fl) + 3.5 * attn_fl   # backward: dgrad + wgrad per GEMM; attention backward = 5 of the forward's 2 products
            train_unfrozen = {"value": round(world * nsu / elu, 3), "unit": "train steps/s (whole job)", "ms_per_step": round(1e3 * elu / nsu, 3),
                              "batch_per_gpu": Bu, "global_batch": Bu * world, "tokens_per_sample": Ni_u + T, "trainable_params": int(total_u),
                              "grad_bytes": int(total_u * 4), "buckets": nb_u, "collectives_per_step": len(bex.launched), "parallelism": f"dp{world}",
                              "exchange": exch_u,
                              "algorithmic_tflop_per_step": round(step_fl / 1e12, 2),
                              "roofline": {"bound": "mfma", "achieved": round(step_fl / (elu / nsu) / 1e12, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": round(step_fl / (elu / nsu) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                           "note": "whole step incl. the frozen tower; the forward's split-bf16 operands execute 2x the algorithmic MFMA work of each projection"},
                              "workspace_gb": round(ws_u.numel() / 2 ** 30, 2),
                              "ms_per_step_fp16_forward": round(1e3 * elu_f16 / nsu, 3),
                              "ms_per_step_split_bf16_dgrad": round(1e3 * elu_bf / nsu, 3),
                              "ms_per_step_split_bf16_dgrad_and_two_pass_bf16_wgrad": round(1e3 * elu_w2 / nsu, 3),
                              "backward_arithmetic": "dgrad and wgrad each ONE fp16 pass (gradient x 2^12 loss scale; worst per-tensor gradient 8.4e-4 from fp32 autograd through all 24 layers; gate/up accumulators kept as fp16)"}
            # leave the engine as it was: the legs below run on the original weights
            eng.train_commit(flat_u0)
            torch.cuda.synchronize()
            del ws_u, g_u, m_u, v_u, flat_u, flat_u0
        except Exception as exc:  # the leg must never take the headline line down with it
            train_unfrozen = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.empty_cache()

    # ---- EVERYTHING trainable: the FastViT-HD tower too (the tower half; fv_train_tower_*).  One step = letterbox + tower forward with every
    # unit's tensors kept + the unfrozen step above + the tower's backward (recomputed ConvFFN hidden, fp16 dgrad / TN wgrad GEMMs, depthwise / attention / norm
    # backward kernels) + clip + AdamW over 494 M + 125 M parameters + refresh of every packed operand image.  Algorithmic flops: 3 x (tower + projector + decoder
    # GEMMs) + attention fwd/bwd (the recomputed fc1 of every ConvFFN is executed, not counted).
    run_tower_leg = train_unfrozen is not None and "error" not in train_unfrozen and not args.no_train_tower
    if world > 1 and not args.no_train_unfrozen and not args.no_train_unfrozen_dp:
        run_tower_leg = ranks_agree(run_tower_leg)      # the previous leg may have failed on one rank only
    if run_tower_leg:
        try:
            setup_err = None
            try:
                eng.train_tower_begin()
                tensors_t, total_t, nb_t = eng.train_layout()
                flat_t = torch.zeros(total_t, dtype=torch.float32, device=dev)
                eng.train_export_params(flat_t)
                flat_t[: flat.numel()].copy_(flat)
                flat_t0 = flat_t.clone()
                g_t, m_t, v_t = torch.zeros_like(flat_t), torch.zeros_like(flat_t), torch.zeros_like(flat_t)
                ws_t, tws_t = eng.train_workspace(Bu, T), eng.train_tower_workspace(Bu)
                dto_t = torch.zeros(Bu, model.tower.num_tokens, model.tower.out_dim, dtype=torch.float16, device=dev)
                eng.train_set_tower_grad(dto_t)
                bex_t = BucketedGradExchange(dev, min_numel=1 << 22)
            except Exception as exc_:
                setup_err = exc_
            if not ranks_agree(setup_err is None):
                raise RuntimeError(f"set-up failed on {'this' if setup_err is not None else 'another'} rank: {setup_err!r}")
            tst = {"step": 0}

            def step_tower():
                tst["step"] += 1
                pix_t = eng.preprocess(images[:Bu])
                tower_out_t = eng.train_tower_forward(pix_t, tws_t)
                bex_t.begin(g_t)
                eng.train_forward_backward(flat_t, tower_out_t, ids[:Bu], lens[:Bu], states[:Bu], targets[:Bu], ws_t, training=True, dropout_p=0.1,
                                           seed=args.seed + rank, offset=tst["step"], flat_grads=g_t, bucket_cb=bex_t.bucket_ready)
                eng.train_tower_backward(pix_t, dto_t, tws_t, g_t, bucket_cb=bex_t.bucket_ready)
                scale = bex_t.finish(dev)
                eng.adamw_step(flat_t, g_t, m_t, v_t, tst["step"], lr=1e-5, weight_decay=1e-4, max_grad_norm=1.0, grad_scale=scale / eng.train_loss_scale())
                eng.train_commit(flat_t)

            nstw = max(2, args.steps // 4)
            eltw = timed(step_tower, nstw, 2)
            exch_t = None
            if world > 1:
                bex_t.dry_run = True
                eltw_dry = timed(step_tower, nstw, 1)
                bex_t.dry_run = False
                exch_t = {"ms_per_step_without_exchange": round(1e3 * eltw_dry / nstw, 3), "exchange_exposed_ms": round(1e3 * (eltw - eltw_dry) / nstw, 3)}
            sat_t = eng.fp16_saturations()
            # where the step goes: one more step under the per-launch event profiler
            eng.profile(True)
            step_tower()
            fams_t, _ = eng.profile_read()
            eng.profile(False)
            step_fl_t = 3.0 * tower_fl + 3.0 * (dec_fl + proj_fl) + 3.5 * attn_fl
            train_unfrozen_tower = {"value": round(world * nstw / eltw, 3), "unit": "train steps/s (whole job)", "ms_per_step": round(1e3 * eltw / nstw, 3),
                                    "batch_per_gpu": Bu, "global_batch": Bu * world, "trainable_params": int(total_t), "tower_params": int(total_t - total_u),
                                    "grad_bytes": int(total_t * 4), "buckets": nb_t, "collectives_per_step": len(bex_t.launched), "parallelism": f"dp{world}",
                                    "exchange": exch_t,
                                    "algorithmic_tflop_per_step": round(step_fl_t / 1e12, 2),
                                    "roofline": {"bound": "mfma", "achieved": round(step_fl_t / (eltw / nstw) / 1e12, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                 "frac": round(step_fl_t / (eltw / nstw) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                                 "note": "whole step; forward + backward of tower, projector, decoder, head; every ConvFFN's fc1 is executed twice (recomputed in the backward)"},
                                    "workspace_gb": round((ws_t.numel() + tws_t.numel()) / 2 ** 30, 2), "fp16_saturations": int(sat_t),
                                    "profiled_step_ms_by_family": {k: round(v["ms"], 3) for k, v in fams_t.items() if v["launches"]},
                                    "backward_arithmetic": "fp16 gradient stream (x 2^12), fp16 dgrad / TN wgrad GEMMs with fp32 accumulation, fp32 tap / norm / layer-scale gradients by fixed-order partial sums"}
            eng.train_set_tower_grad(None)
            eng.train_commit(flat_t0)
            torch.cuda.synchronize()
            del ws_t, tws_t, g_t, m_t, v_t, flat_t, flat_t0, dto_t
        except Exception as exc:  # the leg must never take the headline line down with it
            train_unfrozen_tower = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.empty_cache()

    # ---- the same step through the reference's plugin surface (reference lerobot_fastvla/modeling_fastvla.py:109-133):
    # FastVLAPolicy.select_action(batch) / .forward(batch) with a LeRobot batch dict and B task strings through the tokenizer;
    # Python glue, tokenisation, the action deque and forward()'s `.item()` are inside these numbers.  The policy drives the SAME
    # engine (frozen weights are not loaded twice); its head parameters are its own nn.Parameters.
    surface = None
    if not args.no_surface and not args.splice and world == 1:
        from vla_fastvlm.lerobot_fastvla import FastVLAConfig as LRConfig, FastVLAPolicy as LRPolicy
        from vla_fastvlm.lerobot_fastvla._lerobot_compat import FeatureType, PolicyFeature
        feats = {"observation.images.top": PolicyFeature(FeatureType.VISUAL, (3, args.image, args.image)),
                 "observation.state": PolicyFeature(FeatureType.STATE, (14,))}
        torch.manual_seed(4321)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):   # the backbone announces its image size on stdout like the reference; stdout is the JSON line's
            pol = LRPolicy(LRConfig(vlm_model_name=f"synthetic:{args.model}:{args.seed}", input_features=feats, tokenizer_max_length=T,
                                    output_features={"action": PolicyFeature(FeatureType.ACTION, (14,))}))
        pol.model.backbone._engine = eng          # share the loaded engine (same head dims: 14 / 14 / 1024 / 1024)
        pol.to(dev)
        pol.model.materialize(dev)
        # T-token prompts: the synthetic tokenizer maps one byte to one id, so T - 1 characters + the trailing newline
        tasks = [(f"pick up object {i:03d} and place it in the bin on the left side of the table, then return home " * 2)[: T - 1] for i in range(B)]
        lbatch = {"observation.images.top": images[:B], "observation.state": states[:B], "action": targets[:B, None], "task": tasks}

        def surf_select():
            pol.reset()
            return pol.select_action(lbatch)

        def surf_forward():
            return pol.forward(lbatch)[0]

        def host_ms(fn, n=3):   # time spent INSIDE the call (enqueue + Python), GPU left to run behind it
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            h = (time.perf_counter() - t) / n
            torch.cuda.synchronize()
            return 1e3 * h

        ns = max(3, args.steps // 2)
        sel = timed(surf_select, ns, 2)
        sel_host = host_ms(surf_select)
        pol.train()
        fwd = timed(surf_forward, ns, 2)
        pol.eval()
        eng_ms = ms_per_step
        surface = {"select_action_ms_per_step": round(1e3 * sel / ns, 3), "select_action_host_ms_per_call": round(sel_host, 3),
                   "select_action_vs_engine_level": round((1e3 * sel / ns) / eng_ms, 4),
                   "forward_ms_per_step": round(1e3 * fwd / ns, 3), "steps": ns,
                   "forward_note": "forward(batch) -> (loss, {'loss','mse'}): head forward + MSE + head gradients in train mode, and the "
                                   "reference API's loss.item() host sync every step",
                   "batch": B, "prompt_tokens": T, "tokenizer": type(pol.model.backbone.tokenizer).__name__,
                   "api": "vla_fastvlm.lerobot_fastvla.FastVLAPolicy.select_action(batch) / .forward(batch)"}
        # ---- the control loop the reference's eval scripts actually run (reference lerobot_fastvla/modeling_fastvla.py:119-125: one observation in, one
        # action out, the host waiting for it): select_action on B = 1 / 2 / 4 observations, synchronised every call.  The launchers take the few-row
        # forms here (K ranges for the decoder's 64-row GEMMs, row segments of the depthwise march, hidden ranges of the fused ConvFFN).
        control = {}
        for Bc in (1, 2, 4):
            cb = {"observation.images.top": images[:Bc], "observation.state": states[:Bc], "task": tasks[:Bc]}
            ts = []
            for it in range(25):
                torch.cuda.synchronize()
                t0c = time.perf_counter()
                pol.reset()
                a_c = pol.select_action(cb)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0c)
            ts = sorted(ts[5:])
            ids_c, lens_c = ids[:Bc], lens[:Bc]
            te = []
            for it in range(25):
                torch.cuda.synchronize()
                t0c = time.perf_counter()
                eng.head_forward(flat, eng.backbone(images[:Bc], ids_c, lens_c), states[:Bc])
                torch.cuda.synchronize()
                te.append(time.perf_counter() - t0c)
            te = sorted(te[5:])
            control[f"b{Bc}"] = {"select_action_ms_median": round(1e3 * ts[len(ts) // 2], 3), "select_action_ms_min": round(1e3 * ts[0], 3),
                                 "engine_level_ms_median": round(1e3 * te[len(te) // 2], 3)}
        surface["control_loop"] = dict(control, note="one call = tokenizer + letterbox + tower + projector + decoder + pool + head, host-synchronised; round 4: 5.6 / 6.2 / "
                                                      "6.6 ms at the engine level (tools/latency_small_batch.py)")
        del pol

    # ---- the other decoder parity mode on a second engine (same weights, inputs, head): the opt-in policy 2 trades the decoder's 1e-5
    # for ~5e-4 (batch rel-L2) on the actions; both numbers belong in one line
    alt = None
    if rank == 0 and world == 1 and not args.no_alt and w is not None and args.llm_precision in (1, 2, 5) and not args.splice:
        ap_ = {1: 5, 5: 1, 2: 1}[args.llm_precision]
        eng2 = FastVLAEngine(model, state_dim=14, action_dim=14, hidden_dim=1024, fusion_dim=1024, device=dev, max_batch=B, max_text_tokens=T,
                             tower_microbatch=args.microbatch, llm_precision=ap_)
        eng2.load_weights(w)

        def step_alt():
            pooled = eng2.backbone(images[:B], ids[:B], lens[:B])
            return eng2.head_forward(flat, pooled, states[:B])[0]

        ela = timed(step_alt, args.steps, args.warmup)
        a_main, a_alt = step_infer(), step_alt()
        alt = {"llm_precision": ap_, "ms_per_step": round(1e3 * ela / args.steps, 3), "value": round(args.steps / ela, 4),
               "actions_rel_l2_vs_default_mode": float((a_alt - a_main).norm() / a_main.norm())}
        holder_alt = a_alt[: args.cpu_sample].cpu()
        eng2.close()
        del eng2

    # ---- splice mode only: what the image-prefix cache changes.  joint = ONE prefill over 256 image + T text
    # positions (the timed step above); new frames = tower + prefix pass (image positions, K / V kept) + suffix pass (text
    # positions); cached frames = the suffix pass alone (a repeated frame, or another prompt on the same frame)
    prefix = None
    if args.splice and args.llm_precision >= 1 and model.llm.head_dim >= 64:
        holder = {}

        def step_new_frames():
            tok = eng.vision_forward(eng.preprocess(images[:B]))
            holder["kv"] = eng.llm_prefix(tok)
            act, _ = eng.head_forward(flat, eng.llm_pooled_prefixed(ids[:B], lens[:B], holder["kv"]), states[:B])
            return act

        def step_cached_frames():
            act, _ = eng.head_forward(flat, eng.llm_pooled_prefixed(ids[:B], lens[:B], holder["kv"]), states[:B])
            return act

        ns = max(3, args.steps // 2)
        t_new = timed(step_new_frames, ns, 2)
        t_hit = timed(step_cached_frames, ns, 2)
        prefix = {"joint_prefill_ms_per_step": round(ms_per_step, 3), "new_frames_ms_per_step": round(1e3 * t_new / ns, 3),
                  "cached_frames_ms_per_step": round(1e3 * t_hit / ns, 3), "steps": ns,
                  "cache_mib_per_image": round(holder["kv"].numel() * 4 / B / 2 ** 20, 2),
                  "actions_rel_l2_vs_joint": float((step_cached_frames() - step_infer()).norm() / step_infer().norm())}

    # ---- CPU baseline (rank 0, N=1 only): the fp32 oracle on a bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and w is not None:
        from oracle import fastvit_hd, head as ohead, policy, qwen2
        n = args.cpu_sample
        lc = qwen2.Qwen2Cfg(hidden=model.llm.hidden, layers=model.llm.layers, heads=model.llm.heads, kv_heads=model.llm.kv_heads,
                            head_dim=model.llm.head_dim, inter=model.llm.inter, vocab=model.llm.vocab)
        tc = fastvit_hd.TowerCfg(layers=model.tower.layers, dims=model.tower.dims)
        hp = {k: v.detach().cpu().clone() for k, v in eng.head_views(flat).items()}
        # the box exposes every host core but grants a 16-CPU share per GPU: size the pool to the share
        torch.set_num_threads(_usable_cpus())
        ci, cids = images[:n].cpu(), ids[:n].cpu().long()
        cmask = torch.ones(n, T, dtype=torch.long)
        t = time.perf_counter()
        with torch.no_grad():
            ref = policy.policy_forward(w, hp, ci, states[:n].cpu(), cids, cmask, image_size=model.tower.image_size,
                                        llm_cfg=lc, tower_cfg=tc, splice=args.splice)
        cel = time.perf_counter() - t
        got = step_infer()[:n].cpu()
        err = float((got - ref).norm() / ref.norm())
        if alt is not None:
            alt["parity_actions_rel_l2"] = float((holder_alt[:n] - ref).norm() / ref.norm())
        cpu = {"value": round((n / cel) / B, 5), "unit": f"steps/s (bs={B} equivalent)", "cores": torch.get_num_threads(),
               "kind": "port", "samples_per_s": round(n / cel, 3), "seconds": round(cel, 2),
               "sample": f"{n} images {args.image}x{args.image} + {T}-token prompts, one forward "
                         f"(letterbox+tower+projector+decoder+pool+head) of the fp32 torch oracle",
               "parity_actions_rel_l2": err}

    # ---- C1 protocol: FastVLM-0.5B, bs=4, 336^2 + 32-token prompt, train steps
    # (forward + MSE + head backward + clip + AdamW) of the CPU port, first step excluded, mean +- std; "algorithmic" = the work
    # the result needs, "as_shipped" adds what the reference also computes and drops (full-vocabulary lm_head logits for every
    # token, 25 retained hidden-state tensors, a KV cache).  The GPU path's own C1 step is timed beside it.
    c1 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_c1 and w is not None:
        from oracle import fastvit_hd, head as ohead, policy, qwen2
        import statistics
        Bc, Tc = 4, 32
        lc = qwen2.Qwen2Cfg(hidden=model.llm.hidden, layers=model.llm.layers, heads=model.llm.heads, kv_heads=model.llm.kv_heads,
                            head_dim=model.llm.head_dim, inter=model.llm.inter, vocab=model.llm.vocab)
        tc = fastvit_hd.TowerCfg(layers=model.tower.layers, dims=model.tower.dims)
        torch.set_num_threads(_usable_cpus())
        hp = {k: v.detach().cpu().clone() for k, v in eng.head_views(flat).items()}
        mo = {k: torch.zeros_like(v) for k, v in hp.items()}
        vo = {k: torch.zeros_like(v) for k, v in hp.items()}
        ci, cids = images[:Bc].cpu(), ids[:Bc, :Tc].cpu().long()
        cmask = torch.ones(Bc, Tc, dtype=torch.long)
        cst, ctg = states[:Bc].cpu(), targets[:Bc].cpu()
        emb_w = w["model.embed_tokens.weight"]
        # inputs of the reference-only work, computed ONCE outside the timed steps (the backbone is frozen, they do not change):
        # the letterboxed batch and the decoder's real final hidden states with its 25 retained states
        from oracle import preprocess as opre
        with torch.no_grad():
            pix_c1 = opre.letterbox(ci, model.tower.image_size)
            kept_c1 = {}
            hid_c1 = qwen2.decoder_forward(w, torch.nn.functional.embedding(cids, emb_w), cmask.sum(1), lc, taps=kept_c1)
        alg, shipped, t_start, first_pred = [], [], time.perf_counter(), None
        for i in range(args.c1_steps):
            if i > 1 and time.perf_counter() - t_start > args.c1_budget:
                break
            t = time.perf_counter()
            with torch.no_grad():
                r = policy.train_step(w, hp, mo, vo, i + 1, ci, cst, ctg, cids, cmask, lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0,
                                      image_size=model.tower.image_size, llm_cfg=lc, tower_cfg=tc, splice=args.splice)
            t_alg = time.perf_counter() - t
            if first_pred is None:
                first_pred = r["pred"]
            hp, mo, vo = r["params"], r["m"], r["v"]
            t = time.perf_counter()
            with torch.no_grad():
                # what the reference computes on top, restated: (1) FastVLAProcessor.prepare_images letterboxes to 1024^2 and
                # FastVLMBackbone.forward letterboxes its result AGAIN (fastvla/processor_fastvla.py:35 -> fastvlm_adapter.py:513,
                # 479-488): a second, identity-size bilinear pass over B x 3 x 1024^2; (2) lm_head logits over the full vocabulary
                # for every token of the REAL final hidden states (tied embedding; fastvlm_adapter.py:533 calls the CausalLM);
                # (3) output_hidden_states=True / use_cache=True keep 25 states and the K/V of every layer alive -- references to
                # tensors the forward made anyway: memory, no arithmetic (kept_c1 stands for them)
                again = opre.letterbox(pix_c1, model.tower.image_size)
                logits = torch.nn.functional.linear(hid_c1, emb_w)
                del again, logits
            t_extra = time.perf_counter() - t
            if i > 0:
                alg.append(t_alg)
                shipped.append(t_alg + t_extra)
        # the same C1 step on the GPU path (first step excluded, 10 timed)
        fl = flat.clone()
        mb_, vb_, gb_ = torch.zeros_like(fl), torch.zeros_like(fl), torch.zeros_like(fl)
        sv = eng.head_saved(Bc)
        i32 = ids[:Bc, :Tc].contiguous()
        l32 = torch.full((Bc,), Tc, dtype=torch.int32, device=dev)
        c1s = {"n": 0}

        def c1_step():
            c1s["n"] += 1
            pooled = eng.backbone(images[:Bc], i32, l32, splice=args.splice)
            act, _ = eng.head_forward(fl, pooled, states[:Bc], saved=sv)
            eng.head_backward(fl, act, targets[:Bc], sv, flat_grads=gb_)
            eng.adamw_step(fl, gb_, mb_, vb_, c1s["n"], lr=1e-4, weight_decay=1e-4, max_grad_norm=1.0)
            return act

        g_first = c1_step().cpu()
        g_el = timed(c1_step, 10, 0)
        def ms(v):
            return {"mean_ms": round(1e3 * statistics.mean(v), 1), "std_ms": round(1e3 * (statistics.pstdev(v) if len(v) > 1 else 0.0), 1), "steps": len(v)}
        c1 = {"config": "C1: FastVLM-0.5B bs=4, 336x336 + 32-token prompt, train step (fwd + MSE + head bwd + clip + AdamW), first step excluded",
              "cpu_algorithmic": ms(alg), "cpu_as_shipped": ms(shipped), "cpu_kind": "port (fp32 torch oracle)",
              "cpu_as_shipped_adds": "second identity-size letterbox of the 1024^2 batch + full-vocabulary lm_head logits on the real "
                                     "final hidden states (retained states / KV cache cost memory only)",
              "cpu_threads": torch.get_num_threads(), "os_cpu_count": os.cpu_count(), "steps_requested": args.c1_steps,
              "budget_s": args.c1_budget, "gpu_ms_per_step": round(1e3 * g_el / 10, 3),
              "gpu_vs_cpu_algorithmic": round(statistics.mean(alg) / (g_el / 10), 1),
              "parity_actions_rel_l2_step1": float((g_first - first_pred).norm() / first_pred.norm())}

    # ---- N > 1 on RCCL: the library's OWN communicator (fv_comm_* through the C ABI) -- unique id from rank 0, init on every rank, one all-reduce of ones
    # through fv_allreduce_grads.  Last thing of the run, on a helper thread with a time limit: a failure is recorded, a stall is recorded and the process
    # leaves without joining it; neither costs the line.
    stalled = False
    if world > 1 and dist_info is not None and not args.no_fv_comm_check:
        if args.backend != "nccl" or ndev < world:
            dist_info["fv_comm"] = {"ok": None, "skipped": f"backend {args.backend}, {ndev} device(s) for {world} ranks: RCCL needs one GPU per rank"}
        else:
            res = {}

            def fv_comm_live_check():
                try:
                    box = [eng.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    t0_ = time.perf_counter()
                    comm = eng.comm_init(box[0], rank, world)
                    res["init_ms"] = round(1e3 * (time.perf_counter() - t0_), 1)
                    probe = torch.ones(1 << 20, dtype=torch.float32, device=dev)
                    eng.allreduce_grads(comm, probe)
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    for _ in range(5):
                        eng.allreduce_grads(comm, probe)
                    torch.cuda.synchronize()
                    res["allreduce_4mib_ms"] = round(1e3 * (time.perf_counter() - t0_) / 5, 3)
                    res.update(ranks=world, allreduce_of_ones=float(probe[0]), ok=bool(float(probe[0]) == float(world) ** 6 and bool((probe == probe[0]).all())))
                    eng.comm_destroy(comm)
                except Exception as exc_:
                    res.update(ok=False, error=f"{type(exc_).__name__}: {exc_}")

            th = threading.Thread(target=fv_comm_live_check, daemon=True)
            th.start()
            th.join(timeout=90.0)
            stalled = th.is_alive()
            dist_info["fv_comm"] = {"ok": False, "error": "no answer within 90 s (stalled in fv_comm_init / fv_allreduce_grads)", **res} if stalled else res
    if watchdog is not None:
        watchdog.cancel()
    emit()
    if stalled:
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
