"""bench/main.py — the driver contract of bench.py: argument parsing, the timed region (barrier + device sync on both sides, MAX over ranks),
the per-kernel HIP-event pass, roofline / cpu_baseline / parity blocks, and ONE JSON line on rank 0."""
from .common import *  # noqa: F401,F403
from .wl_csv import CsvWorkload
from .wl_messages import JsonWorkload, SrWorkload
from .wl_configs2 import Configs2Workload
from .wl_configs3 import Configs3Workload
from .wl_configs0 import Configs0Workload
from .wl_configs4 import Configs4Workload, Configs4DebeziumWorkload, CollapseWorkload
from .wl_debezium import DebeziumWorkload, DebeziumSrWorkload, SrProtoWorkload

WORKLOADS = {"configs0": Configs0Workload, "debezium": DebeziumWorkload, "debezium_sr": DebeziumSrWorkload, "sr_proto": SrProtoWorkload, "configs4d": Configs4DebeziumWorkload, "csv": CsvWorkload, "json": JsonWorkload, "sr": SrWorkload, "configs2": Configs2Workload, "configs3": Configs3Workload,
             "configs4": Configs4Workload, "collapse": CollapseWorkload}


class Env:
    pass


# BASELINE.json's other configurations (and the Debezium parser, the source half of configs[4]) on the default line: each runs as its OWN short bench
# process after the headline's timed region — its own context, its own cpu_baseline / parity leg — and only a digest of its line is kept here.
OTHER_CONFIGS = [("configs0", "configs[0]", []), ("configs2", "configs[2]", []), ("configs3", "configs[3]", []), ("configs4", "configs[4]", ["--overlap-lanes", "3"]), ("debezium", "configs[4] source format", ["--overlap-lanes", "3"])]


def other_configs(args):
    import subprocess
    out = {}
    for w, which, extra in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--gpus", "1", "--steps", "8", "--warmup", "2", "--pcie-steps", "0",
               "--cpu-rows", str(1 << 16), "--cpu-all-rows", "0", "--parity-rows", str(args.parity_rows), "--other-configs", "0"] + (extra or ["--overlap-lanes", "0"])
        t0 = time.perf_counter()
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
            d = None
            for ln in reversed(r.stdout.strip().splitlines()):
                try:
                    d = json.loads(ln)
                    break
                except ValueError:
                    continue
            if r.returncode != 0 or d is None:
                out[w] = {"baseline_config": which, "error": (r.stderr or r.stdout)[-300:]}
                continue
            rf, par, cpu = d.get("roofline") or {}, d.get("parity") or {}, d.get("cpu_baseline") or {}
            if not rf.get("kernel") and d.get("kernels"):  # a step of many small kernels none of which is priced in bytes (configs[4]): the longest is still named
                kn, kv = max(d["kernels"].items(), key=lambda kv: kv[1].get("ms_per_step", 0))
                rf = {"kernel": kn, "avg_launch_ms": kv.get("avg_ms"), "note": "not priced against the HBM peak (no algorithmic byte count for it)"}
            frac = rf.get("frac")
            ent = {"baseline_config": which, "workload": d["config"].get("workload", "")[:160], "value": d["value"], "unit": d["unit"], "ms_per_pass": round(d["ms_per_step"] / max(d.get("passes_per_step", 1), 1), 4),
                   "rows_per_pass": d["config"].get("rows_per_pass"), "dominant_kernel": rf.get("kernel"), "dominant_kernel_ms": rf.get("avg_launch_ms"), "frac": frac,
                   "achieved_gb_s": rf.get("achieved"), "kernels_consistent_with_timed_region": rf.get("consistent"),
                   "parity": {"identical": bool(par.get("identical")), "checked": par.get("checked_input_rows", 0), "what": (par.get("checked") or par.get("error") or "")[:240]},
                   "cpu_port_rows_per_s": cpu.get("value"), "seconds": round(time.perf_counter() - t0, 1)}
            ol = d.get("overlapped_lanes") or {}
            if ol.get("rows_per_s"):  # never `value`: the same steps on three lanes (Base.overlapped_lanes) — what the one-lane figure leaves idle between its kernels, filled
                ent["three_lanes_rows_per_s"] = ol["rows_per_s"]
            if frac is not None and frac > 1.0:
                ent["frac"], ent["refused"] = None, "a fraction above 1 is not printed"
            out[w] = ent
        except Exception as ex:  # noqa: BLE001
            out[w] = {"baseline_config": which, "error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="csv",
                    help="csv = BASELINE.json configs[1] (the bench line); configs0 = the plumbing case; configs2/3/4 = the other GPU configs; json / sr / collapse = component benchmarks")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: enough for a timed region of >= 1 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU per step (default: 2^20 for csv / configs3 / configs4, 2^18 for the message workloads)")
    ap.add_argument("--cpu-rows", type=int, default=1 << 19, help="rows of the single-thread CPU-baseline sample (0 = skip); 2^19 hits rows = about 11 s of oracle time")
    ap.add_argument("--parity-rows", type=int, default=1 << 16, help="rows / messages of every line's parity leg (the device's output for the oracle's own sample, compared after the timed region)")
    ap.add_argument("--cpu-all-rows", type=int, default=1, help="1 = also time the oracle on every hardware thread (persistent workers, csv workload), 0 = skip")
    ap.add_argument("--prof-steps", type=int, default=10, help="passes of the per-kernel HIP-event measurement (median per kernel)")
    ap.add_argument("--passes", type=int, default=0, help="passes of the path per step (0 = as many as make a step of about --step-ms)")
    ap.add_argument("--step-ms", type=float, default=50.0)
    ap.add_argument("--lanes", type=int, default=1, help="device lanes (host threads) the steps are spread over (1 = strictly serial, the bench line)")
    ap.add_argument("--overlap-lanes", type=int, default=3, help="side measurement (never `value`): the same steps spread over this many lanes (0 = skip)")
    ap.add_argument("--no-pull-push", action="store_true", help="csv workload: skip the configs[2] pull || push side measurement")
    ap.add_argument("--pcie-steps", type=int, default=2, help="side measurement (never `value`): passes per lane that start from PINNED HOST memory (0 = skip)")
    ap.add_argument("--pcie-lanes", type=int, default=3)
    ap.add_argument("--devices", default="", help="ONE process driving several devices (csv workload): comma-separated HIP device ids, lane k on the k-th "
                    "(tfgpu_init_devices); without it --gpus N > 1 expects one process per GPU under torch.distributed.run")
    ap.add_argument("--sink", default="native", choices=["native", "debezium"], help="configs4: the queue sink's format — the native serializer, or the Debezium emitter "
                    "(queue.DebeziumSerializer, key + value with inline schemas: tfgpu_debezium_emit)")
    ap.add_argument("--from-rows", action="store_true", help="configs0: also cross the boundary the way the reference would — boxed []interface{} rows fanned out into column buffers, "
                    "one crossing, fanned back in (tools/fanout/fanout_harness.cpp, INTEGRATION.md §2), timed per leg")
    ap.add_argument("--toast", type=float, default=0.0, help="collapse workload: this share of the Updates leaves the text column out of its ColumnNames (TOAST): Collapse runs the compareColumns merge")
    ap.add_argument("--other-configs", type=int, default=1, help="csv workload at N=1: 1 = after the headline's own measurements, run configs0 / 2 / 3 / 4 and debezium as short benches of their own and "
                    "carry a digest of each line (rows/s, ms per pass, dominant kernel + fraction of the HBM peak, parity) as `other_configs`; 0 = skip")
    ap.add_argument("--exchange", type=int, default=1, help="configs4 at 1 rank: 1 = every buffer still makes the RCCL round trip, 0 = skip the collective")
    args = ap.parse_args()

    import torch  # first: libtfgpu and torch must share the HIP runtime torch loads
    from transferia_amd import abi, dist as tdist, lib, workload
    if os.environ.get("TFGPU_LIB_VARIANT"):  # measurement only: an A/B build of the same sources (tools/build_variant.sh); the binding itself knows one library
        lib._LIBPATH = os.path.join(os.path.dirname(lib._LIBPATH), "variants", "libtfgpu_%s.so" % os.environ["TFGPU_LIB_VARIANT"])
    rank, local_rank, world = tdist.env_rank()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # TEST INFRASTRUCTURE: tests/test_hipemu.py runs `bench.py --devices 0,1` end to end on the CPU emulator's fake devices (tools/hipemu, HIPEMU_DEVICES=2) so
    # that the one-process-several-GPUs path cannot rot on one-GPU boxes; the numbers of such a run mean nothing and its line says so.
    emu = os.environ.get("TFGPU_TEST_EMU_LIB")
    if emu:
        lib._LIBPATH = emu
    else:
        torch.cuda.set_device(local_rank)
    e = Env()
    e.rank, e.local_rank, e.world = rank, local_rank, world
    e.device = torch.device("cpu") if emu else torch.device("cuda", local_rank)
    e.group = tdist.Group("gloo" if emu else "nccl", e.device)  # RCCL: barrier + MAX of the wall time (+ the configs4 exchange)
    e.lib, e.abi, e.workload = lib, abi, workload
    e.devices = [int(x) for x in args.devices.split(",") if x.strip() != ""] if args.devices else None
    if e.devices:
        if world != 1 or args.workload != "csv":
            raise SystemExit("--devices is the one-process mode of the csv workload: run it without torch.distributed.run")
        lib.init_devices(e.devices)
        args.overlap_lanes = 0; args.pcie_steps = 0
    else:
        lib.init(local_rank)

    def sync_all():
        lib.synchronize()
        e.group.barrier()
        if not emu:
            torch.cuda.synchronize()
    e.sync_all = sync_all

    W = WORKLOADS[args.workload](args, e)
    if args.rows <= 0:
        args.rows = W.default_rows
    W.setup()
    for _ in range(args.warmup):
        W.step()
    W.step(keep=True)  # one extra untimed pass records sizes
    # One *step* of the headline loop is `passes` passes of the path (each over the whole resident batch): a pass takes ~1.5 ms,
    # and a driver that asks for --steps 20 would otherwise time a 30 ms region no utilisation sampler can see.
    passes = args.passes
    if passes <= 0:
        lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            W.step()
        lib.synchronize()
        per = (time.perf_counter() - t0) / 3
        passes = max(1, int(np.ceil(args.step_ms * 1e-3 / max(per, 1e-6))))
        if e.group.dist is not None:  # every rank must time the same number of passes
            t = torch.tensor([passes], dtype=torch.int64, device=e.device)
            e.group.dist.all_reduce(t, op=e.group.dist.ReduceOp.MAX)
            passes = int(t.item())
    if args.steps <= 0:  # no K given: a timed region of >= 1 s
        args.steps = 24
    W.total_passes = args.steps * passes
    if os.environ.get("TFGPU_BENCH_HOSTPROF") == "1":  # where the HOST spends a pass (stderr; the JSON line is unaffected): cProfile over a few untimed passes
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            W.step()
        lib.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
    my_dt = W.timed(args.steps * passes)
    dt = e.group.max_seconds(my_dt)
    W.dt = dt
    # the N-GPU run validates itself: every rank's own time and rows arrive on rank 0 through the collective library
    rank_ms = e.group.all_gather_float(my_dt / args.steps * 1e3)
    rank_rows = e.group.all_gather_float(float(W.rows() * passes))
    ranks_seen = e.group.sum_int(1)
    ms_per_pass = dt / (args.steps * passes) * 1e3

    # ---- per-kernel device time: HIP events on the library stream, IMMEDIATELY after the timed region (before any side
    #      measurement touches clocks, lanes or PCIe), 2 warm-up passes, then >= 10 passes read back one by one: the figure kept
    #      per kernel is the MEDIAN over passes of (sum of its launches in the pass) ----
    lib.prof_reset(); lib.prof_enable(True)
    for _ in range(2):
        W.step()
    lib.prof_reset()
    per_pass, units = {}, {}
    nprof = max(args.prof_steps, 1)
    for i in range(nprof):
        W.step()
        for n, l, ms in lib.prof_get():
            if l:
                per_pass.setdefault(n, []).append((l, ms))
        for n, u in lib.prof_units().items():
            if u:
                units.setdefault(n, []).append(u)
        lib.prof_reset()
    lib.prof_enable(False)
    kernels = {}
    for n, samples in per_pass.items():
        if len(samples) < nprof:  # not launched in every pass: averaged over all passes
            ms_step = sum(ms for _, ms in samples) / nprof
            lps = sum(l for l, _ in samples) / nprof
        else:
            ms_step = float(np.median([ms for _, ms in samples]))
            lps = float(np.median([l for l, _ in samples]))
        kernels[n] = {"launches_per_step": lps, "avg_ms": ms_step / max(lps, 1e-9), "ms_per_step": ms_step}
        if units.get(n):  # the rows the kernel's launches were issued over, as the library counted them (tfgpu_prof_get_units)
            kernels[n]["units_per_step"] = int(np.median(units[n]))
    kernel_sum_ms = sum(v["ms_per_step"] for v in kernels.values())

    side = W.side_measurements() if hasattr(W, "side_measurements") else {}
    if "overlapped_lanes" not in side and args.overlap_lanes > 1 and world == 1 and not getattr(e, "devices", None) and args.workload not in ("csv", "configs0", "configs2", "configs3"):
        ols = [W.overlapped_lanes(args.overlap_lanes, max(min(args.steps * passes, 60), 6)) for _ in range(int(os.environ.get("TFGPU_BENCH_OL_REPEAT", "3")))]   # the message workloads: see Base.overlapped_lanes
        ols = [o for o in ols if o]
        if ols:
            good = sorted((o for o in ols if "rows_per_s" in o), key=lambda o: o["rows_per_s"])
            ol = dict(good[len(good) // 2]) if good else ols[0]   # the median of the repeats; all of them listed
            if good:
                ol["repeats_rows_per_s"] = [o["rows_per_s"] for o in ols if "rows_per_s" in o]
            side = dict(side); side["overlapped_lanes"] = ol

    W.kernels = kernels
    alg = W.alg()
    dom = max(kernels.items(), key=lambda kv: kv[1]["ms_per_step"])[0] if kernels else None
    roofline = None
    if dom and alg.get(dom) and not emu:   # (the emulator's events carry no time)
        per_launch = alg[dom] / max(kernels[dom]["launches_per_step"], 1)
        # The kernels of a pass cannot take longer than the pass that contains them.  When the event pass says otherwise (another
        # clock state, a noisy box) the dominant kernel is bounded from the TIMED region instead: pass time minus the other kernels.
        consistent = kernel_sum_ms <= 1.05 * ms_per_pass
        launch_ms, source = kernels[dom]["avg_ms"], "hip_events_median"
        if not consistent:
            others = kernel_sum_ms - kernels[dom]["ms_per_step"]
            bound_ms = max(ms_per_pass - others * ms_per_pass / kernel_sum_ms, 1e-6) if others >= ms_per_pass else ms_per_pass - others
            launch_ms, source = bound_ms / max(kernels[dom]["launches_per_step"], 1), "upper bound: timed pass minus the other kernels (event pass inconsistent with the timed region)"
        achieved = per_launch / (launch_ms * 1e-3) / 1e9
        # HBM traffic per launch cannot be read inside this process: it comes from the rocprofv3 --pmc passes of tools/gpu_visit.sh
        # (FETCH_SIZE + WRITE_SIZE with the guide's gfx950 corrections), which stamp profiles/pmc_traffic.json with the sha256 of
        # the kernel's source file — a number measured on another build of the kernel (or another launch shape) is not printed.
        traffic = None
        try:
            import hashlib
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(dom)
            if t and t["rows_per_launch"] == W.rows() and t.get("workload", "csv") == args.workload:
                with open(os.path.join(ROOT, "transferia_amd", "csrc", t["source_file"]), "rb") as f:
                    fresh = hashlib.sha256(f.read()).hexdigest() == t["source_sha256"]
                traffic = t["bytes_per_launch"] if fresh else None
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "algorithmic_bytes_per_launch": int(per_launch), "avg_launch_ms": round(launch_ms, 4), "launch_ms_source": source,
                    "consistent": bool(consistent), "kernels_ms_per_pass": round(kernel_sum_ms, 4), "timed_ms_per_pass": round(ms_per_pass, 4), "event_passes": nprof}
        # `achieved` counts the conservative figure (input bytes + fixed-width values).  The CSV parse kernel also writes a (length, position)
        # pair per text cell — the late-materialised form of SURVEY 8(d)'s "every output column byte", 8 bytes per cell instead of offset + text:
        # stated next to it, never instead of it
        extra = W.alg_views().get(dom) if hasattr(W, "alg_views") else None
        if extra:
            extra = extra / max(kernels[dom]["launches_per_step"], 1)
            roofline["text_view_bytes_per_launch"] = int(extra)
            roofline["frac_with_text_views"] = round((per_launch + extra) / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    int_roof = W.int_roofline(kernels) if not emu else None
    for k, v in kernels.items():
        b = alg.get(k)
        if b and v["avg_ms"] > 0:
            v["alg_gb_s"] = round(b / max(v["launches_per_step"], 1) / (v["avg_ms"] * 1e-3) / 1e9, 2)
        v["avg_ms"], v["ms_per_step"] = round(v["avg_ms"], 4), round(v["ms_per_step"], 4)

    cpu = W.cpu() if (rank == 0 and world == 1 and args.cpu_rows > 0) else None  # the oracle, rank 0, N = 1 only

    if rank == 0:
        rows = W.rows()
        shards = len(e.devices) if e.devices else world
        value = rows * shards * args.steps * passes / dt
        out = {"metric": W.metric, "value": round(value, 1), "unit": "rows/s", "n_gpus": len(set(e.devices)) if e.devices else world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": W.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": W.config()}
        out["passes_per_step"] = passes
        out["config"]["rows_per_pass"] = rows
        out["config"]["rows_per_gpu_per_step"] = rows * passes
        for key in [k for k in out["config"] if k.endswith("_bytes_per_gpu_per_step")]:  # setup() knows one pass; a step is `passes` of them
            out["config"][key.replace("_per_gpu_per_step", "_per_pass")] = out["config"][key]
            out["config"][key] = out["config"][key] * passes
        out.update(W.extra())
        if e.devices:
            out["process_model"] = {"mode": "one process, one host thread and one lane per device (tfgpu_init_devices)", "devices": e.devices, "row_range_shards": shards}
        out["multi_gpu"] = {"rccl_ranks_seen": ranks_seen, "world_size": world, "per_rank_ms_per_step": [round(x, 3) for x in rank_ms],
                            "per_rank_rows_per_step": [int(x) for x in rank_rows],
                            "n1_equivalent": "every rank runs exactly the N=1 step over its own row-range shard (rows %d.. of the same synthetic table): value = sum of per_rank_rows_per_step x steps / max time" % (rows * (world - 1)),
                            "measured_beyond_one_gpu": "nothing beyond N=1 had been measured on hardware when this was written (README.md): at N>1 this line is the first evidence" if world > 1 else None}
        if emu:
            out["emulated"] = "tools/hipemu (CPU): a structural run of this command line, not a measurement"
        out["row_errors"] = W.state.get("errors", 0)
        out["roofline"] = roofline
        if int_roof:
            out["int_roofline"] = int_roof
        out["cpu_baseline"] = cpu
        out["kernels"] = kernels
        out.update(side)
        # A fraction above 1 means the arithmetic around a kernel is wrong (or the kernel is not doing the work it is priced for):
        # such a line is not printed.
        bad = [(n, r.get("kernel"), f) for n, r in (("roofline", roofline), ("int_roofline", int_roof)) if r
               for f in (r.get("frac"), r.get("issue_frac"), r.get("frac_with_text_views")) if f is not None and f > 1.0]
        for n, v in kernels.items():  # a secondary kernel priced against bytes it does not see (a hand-over path that ran on a few rows): its figure is withheld, not printed
            if v.get("alg_gb_s", 0) > HBM_PEAK_GBS and n != dom:
                v["alg_gb_s"] = None
                v["alg_note"] = "launched on a fraction of the batch: not priced"
        if bad:
            raise SystemExit("bench.py refuses to print a roofline fraction above 1: %r" % bad)
        if cpu and isinstance(cpu.get("parity"), dict):
            out["parity_checked_rows"] = cpu["parity"].get("checked_input_rows", 0) if cpu["parity"].get("identical") else 0
            out["parity"] = {k: v for k, v in cpu["parity"].items() if k in ("identical", "checked", "checked_input_rows", "error")}
        # the comparison that means something end to end: input from pinned host memory (PCIe inside) against the CPU path on
        # every host core; the HBM-resident `value` over one CPU core is a ratio of two different jobs and is not printed
        pc = (side.get("pcie_inclusive") or {}) if isinstance(side, dict) else {}
        best = max((v.get("rows_per_s", 0) for k, v in pc.items() if isinstance(v, dict)), default=0)
        if cpu and best and isinstance(cpu.get("all_cores"), dict) and cpu["all_cores"].get("value"):
            out["pcie_inclusive_vs_cpu_all_cores"] = round(best / cpu["all_cores"]["value"], 1)
            out["pcie_inclusive_vs_cpu_all_cores_caveat"] = ("the CPU side is a C restatement of the Go reference, by its own note ~2-3x slower per core than the Go binary would be: "
                                                             "against the Go binary read this ratio as roughly a third to a half of what is printed")
        if args.workload == "csv" and world == 1 and not e.devices and args.other_configs and args.cpu_rows > 0 and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
            out["other_configs"] = other_configs(args)
        try:  # RCCL prints its version banner through C stdio: flush that first, the JSON line is the last one
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    e.group.close()
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return  # under rocprofv3 the tool writes its files at normal interpreter exit
    os._exit(0)  # nothing may print after the JSON line

