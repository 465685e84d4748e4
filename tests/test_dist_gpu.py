"""GPU: the sharded pricing path with 2 and 3 ranks (one process each).  Ranks use distinct devices when the box
has that many GPUs (tools/shard_test.py), otherwise they share the one visible GPU — the device mailboxes are
mapped across processes through HIP IPC either way, so the full device-side exchange protocol runs.
Gate: the sharded run picks the identical pivot sequence to the unsharded run."""
import os
import subprocess
import sys

import pytest

from tests.common import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,pivots", [(2, 600), (3, 400)])
def test_sharded_pricing_matches_unsharded_pivot_for_pivot(world, pivots):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), str(world), "4000", "3500", "12", str(pivots)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout


@pytest.mark.parametrize("world,extra", [(2, {}), (3, {}), (2, {"MLP_NO_WSHARD": "1"}), (2, {"MLP_TRANSPORT": "host"})],
                         ids=["2 ranks, row-sharded W stream", "3 ranks, row-sharded W stream", "2 ranks, replicated stream", "2 ranks, host mailbox"])
def test_sharded_pricing_with_the_large_nucleus_machinery(world, extra):
    """Same gate with the delayed-update mode, the strip-shaped streaming pass, the padded pitch of W and the blocked
    F push forced on (they are otherwise used from capacity 8192 on).  With the peer transport the streaming pass over
    the nucleus inverse is ROW-SHARDED: every rank streams the strips s with s % world == rank and the tau_K rows /
    v_K partials are exchanged through the peers' device buffers (k_post_exchange); MLP_NO_WSHARD and the host
    mailbox keep the pass replicated."""
    env = dict(os.environ, MLP_LOWRANK="3", MLP_BIGTILE="1", MLP_LDPAD="16", MLP_BANDED="1", MLP_STR_K="0", **extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), str(world), "4000", "3500", "12", "400"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_dual_loop_matches_unsharded_pivot_for_pivot(world):
    """The dual simplex loop (covering LP: dual feasible, primal infeasible at x = 0) with the three dual
    exchanges: leaving row adopted from rank 0, pass-1 minimum all-reduce, pass-2 candidate all-gather."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), str(world), "3000", "3500", "12", "400", "cover"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout


def test_host_mailbox_transport_still_works():
    """MLP_TRANSPORT=host: the older transport (one mailbox in host memory, polled across PCIe) behind the same protocol."""
    env = dict(os.environ, MLP_TRANSPORT="host")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), "2", "3000", "3000", "12", "300"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout and "host-mapped" in r.stdout


def test_default_transport_is_device_resident():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), "2", "3000", "3000", "12", "300"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout and "device-resident mailboxes" in r.stdout


@pytest.mark.parametrize("world,family", [(2, "sparse"), (3, "cover")])
def test_sharded_pricing_with_the_sparse_tableau_row(world, family):
    """The sparse tableau row (small nucleus) in a sharded solve: every rank lists and pulls the touched columns of its own
    block of non-basic positions; forced here for the whole run (MLP_STR_K), primal and dual loop."""
    env = dict(os.environ, MLP_STR_K="100000", MLP_HYPER="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), str(world), "3000", "3500" if family == "cover" else "2600", "12", "400", family],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout


LATE_BASIS = os.path.join(ROOT, "tests", "golden", "cfg4_basis_p240000.bin.gz")


@pytest.mark.parametrize("world,pivots", [(2, 256), (8, 64)], ids=["2 ranks", "8 ranks (oversubscribed)"])
def test_sharding_at_config4_size_from_the_late_basis(world, pivots):
    """Config 4 itself (100 000 x 100 000), continued from the committed basis with a nucleus of 20 493 and the DEFAULT
    machinery: column-block sharded sweep / update / pricing, ROW-SHARDED streaming pass of the 3.4 GB nucleus inverse
    with the tau_K / v_K exchange through the peers' buffers, replicated fold.  All ranks share the one GPU of the test
    box (so the Harris tests take their two-launch form); the sharded run must take the unsharded run's pivots."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), str(world), "100000", "100000", "100", str(pivots),
                        "basis=" + LATE_BASIS], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout and "device-resident mailboxes" in r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solve_to_optimality_is_optimal_for_a_fresh_unsharded_engine(world):
    """A whole solve under sharding (6 723 pivots, every exchange of the protocol each pivot), then the final partition —
    mlp_solution_save_basis mode 0, the one mode a sharded solution offers — is loaded into a fresh UNSHARDED engine: it
    must be optimal as it stands (no further pivot) and pass the duality certificate (tools/shard_full_solve.py; the same
    tool run on config 4 itself is what exposed the replica-divergence defect of DESIGN.md §6: sharded solves now take
    the deterministic F products so that the ranks stay bit-identical; profiles/r03g_sharded_long_run.log)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_full_solve.py"), str(world), "3000", "3000", "12"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert len({(g["pivots"], g["objective"]) for g in rec["all_ranks"]}) == 1          # every rank took the same pivots
    ru = rec["reloaded_unsharded"]
    assert ru["optimal_as_loaded"] and ru["further_pivots"] == 0
    assert abs(ru["objective"] - rec["all_ranks"][0]["objective"]) <= 1e-9 * abs(ru["objective"])
    ce = ru["certificate"]
    assert ce["relative_gap"] < 1e-9 and ce["max_primal_violation"] < 1e-9 and ce["max_dual_violation"] < 1e-7


@pytest.mark.parametrize("mode", ["default", "replicated"])
def test_sharded_ranks_stay_bit_identical_replicas_over_8000_pivots_from_the_late_basis(mode):
    """The gate that can SEE replica divergence (the 256-pivot identical-pivot gate above could not; the round-3
    defect showed after ~6 000 pivots).  Two ranks continue config 4 from the committed late basis (nucleus 20 493) for 8 000
    pivots with the DEFAULT sharded machinery — deterministic blocked F push (fixed-point limbs), row-sharded streaming pass —
    and stop every 1 000 pivots; tools/shard_bitwise.py compares SHA-1 digests of what every rank holds:
      * at EVERY checkpoint the ranks hold the same bits of x_B, of the basic / non-basic sets and of the nucleus inverse and its
        slot maps (order-independent checksum), and at the last one of the dual steepest-edge weights rebuilt from the inverse;
      * the objective keeps the unsharded run's pace to 1e-9 relative at every checkpoint;
      * mode "replicated" (MLP_NO_WSHARD=1: the streaming pass is not split, so the sharded arithmetic equals the unsharded
        run's operation for operation): the UNION of the ranks' d and gamma blocks, x_B and the inverse equal the unsharded
        run's bit for bit at every checkpoint."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_bitwise.py"), "2", "8000", "1000", mode],
                       capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.loads(lines[-1])
    assert len(rec["checkpoints"]) == 8 and rec["checkpoints"][-1]["pivots"] >= 8000
    for cp in rec["checkpoints"]:
        assert cp["replicas_identical"], (cp["pivots"], cp["replicas_differ_in"])
        assert cp["objective_rel_diff"] <= 1e-9, cp
        if mode == "replicated":
            assert cp["bitwise_equal_to_unsharded"], (cp["pivots"], cp["vs_unsharded"])
    assert r.returncode == 0 and rec["ok"]


@pytest.mark.parametrize("family,rows,cols", [("sparse", "3000", "3000"), ("cover", "3000", "3500")], ids=["primal loop", "dual loop"])
def test_pump_transport_delivers_every_exchange_between_two_ranks(family, rows, cols):
    """MLP_TRANSPORT=pump: the protocol of the RCCL transport — the kernels post into and poll their OWN device box, the host
    pumps stage -> all-gather -> deliver rounds on a second stream until every rank's batch has drained — with peer copies
    between IPC-mapped staging buffers in place of ncclAllGather (RCCL refuses two ranks on one device, which is what this box
    has).  Every exchange kind travels through it: pricing all-gather and ratio decision (primal), leaving row, pass-1 minimum
    and pass-2 candidate (dual); the sharded run must take the unsharded run's pivots."""
    env = dict(os.environ, MLP_TRANSPORT="pump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), "2", rows, cols, "12", "300", family],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout and "transport: pump" in r.stdout


def test_rccl_transport_with_a_world_of_one():
    """MLP_TRANSPORT=rccl end to end with the one rank a one-GPU box allows: librccl.so is loaded at run time, ncclGetUniqueId /
    ncclCommInitRank / ncclAllGather run for real (every batch of pivots is accompanied by pump rounds whose collective is the
    RCCL all-gather), the handshake passes through it, and the solve takes the pivots of a solve without any transport."""
    import minilp_amd as M
    from minilp_amd import api, dist as md, lpgen
    lp = lpgen.gen_sparse_lp(1500, 1400, 12, 9)
    prob = lpgen.build_problem(M.Problem, lp)
    ref = prob.solve(trace=True)
    s = prob.solve(budget=0, trace=True)
    box = md.create_mailbox(1)
    try:
        s.enable_sharding_ex(0, 1, box, "rccl", api.rccl_unique_id())
        assert s.transport().startswith("RCCL")
        s.continue_solve(-1)
    finally:
        md.remove_mailbox(box)
    assert [t[:5] for t in s.trace()] == [t[:5] for t in ref.trace()]
    assert abs(s.objective() - ref.objective()) <= 1e-9 * abs(ref.objective())


def test_rccl_transport_between_two_gpus():
    """The north_star-literal transport with a PEER: two ranks on two devices, every per-pivot record delivered by ncclAllGather.  RCCL
    refuses two ranks on one device, so this runs only where >= 2 GPUs are visible (the first multi-GPU box: tools/first_8gpu_run.sh runs
    the suite first) and is skipped on the one-GPU test box, where `pump` exercises the same protocol with peer copies."""
    import minilp_amd as M
    if M.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, MLP_TRANSPORT="rccl")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), "2", "3000", "3000", "12", "300"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout and "transport: RCCL" in r.stdout


@pytest.mark.slow
def test_compact_factor_under_sharding_takes_the_unsharded_pivots():
    """the compact factor no longer switches off at world > 1 but when ranks SHARE a device (two persistent
    grid-barrier kernels of two processes time-slice the GPU while their peers spin on mailboxes: seconds per pivot).  On this one-GPU box
    that is exactly the situation, so MLP_FACTOR_SHARED_DEVICE=1 keeps the factor for 20 pivots of the transport family — slow, and the
    only way to execute the sharded factor path here: solves replicated on every rank in fixed-order sums, dual ratio test / tableau row /
    update over column blocks; the ranks must take the unsharded factor run's pivots with the factor still active at the end."""
    env = dict(os.environ, MLP_FACTOR="1", MLP_FACTOR_SHARED_DEVICE="1", MLP_SHARD_DEFER="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), "2", "400", "600", "4", "20", "transport"],
                       capture_output=True, text=True, timeout=1400, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout
    assert "compact factor active on the sharded ranks: [1, 1]" in r.stdout and "factor active on rank 0: 1" in r.stdout, r.stdout[-1500:]


def test_sharded_solve_with_the_large_nucleus_machinery_survives_the_polish_step():
    """A whole sharded solve with the delayed-update mode, the row-sharded streaming pass and the blocked push forced on AND the
    polish of long runs forced early (MLP_FINAL_REFRESH: re-inversion, x_B = B^-1 (b - N x_N) by the dense-rhs solve, reduced costs
    recomputed).  The dense solves outside the pivot loop run replicated in the classic tiling; in round 4 they were found to read
    the exchange buffer of the row-sharded pass instead (the 2-rank solve of config 4 ended with an objective of zero on the
    ranks while its final basis was optimal).  Every rank must report the objective a fresh unsharded engine finds for the final basis."""
    import json
    env = dict(os.environ, MLP_FINAL_REFRESH="1000", MLP_LOWRANK="3", MLP_BIGTILE="1", MLP_LDPAD="16", MLP_BANDED="1", MLP_STR_K="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_full_solve.py"), "2", "3000", "3000", "12"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    ru = rec["reloaded_unsharded"]
    assert ru["optimal_as_loaded"] and ru["further_pivots"] == 0
    for g in rec["all_ranks"]:
        assert abs(g["objective"] - ru["objective"]) <= 1e-9 * abs(ru["objective"]), (g, ru["objective"])
    assert ru["certificate"]["relative_gap"] < 1e-9


def _bench(args, env_extra, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MLP_OVERSUBSCRIBE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python3 bench.py --gpus 2 --steps 20 --warmup 5` with no launcher around it must BE a 2-rank run:
    bench.py re-executes itself through torch.distributed.run, one process per GPU (both on the one GPU of this box, which
    MLP_OVERSUBSCRIBE=1 has to allow), and the line says what ran: n_gpus == 2 == ranks_seen, one device index per rank, the
    transport of the per-pivot exchanges, the size of the process group."""
    import json
    r = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-full-solve", "--no-factor-transport"], {"MLP_OVERSUBSCRIBE": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["warmup"] == 5
    rk = rec["ranks"]
    assert rk["ranks_seen"] == 2 and len(rk["devices"]) == 2 and rk["pids_distinct"] and rk["self_launched"]
    assert rk["process_group"]["size"] == 2
    assert rk["transport"] and all("none" != t for t in rk["transport"])
    # the driver's 20 pivots run as replicas (deferred sharding): the line says so instead of calling it strong scaling, and gives no
    # 1-GPU ratio for it; the sharded figures sit at the top level
    assert rec["config"]["completed_steps"] == 20
    assert rk["sharding_live_in_timed_window"] is False and rec["scaling"] == "replicated" and rec.get("value_vs_1gpu") is None
    assert rec["value"] > 0
    late = rec["windows"]["late_sharded"]
    assert "error" not in late and late["us_per_pivot"] > 0
    assert abs(rec["late_sharded"]["us_per_pivot"] - late["us_per_pivot"]) <= 0.11 and rec["late_sharded"]["k"] == 20493
    assert rec.get("pricing_speedup_vs_1gpu") is not None


def test_bench_refuses_more_ranks_than_gpus_without_the_override():
    """... and on a box with ONE GPU the same command without MLP_OVERSUBSCRIBE=1 is an error (rc != 0, no JSON line), never
    a 1-GPU run printed under another name."""
    import minilp_amd as M
    if M.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    r = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-full-solve", "--no-factor-transport"], {}, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


@pytest.mark.parametrize("world,family,rows,cols", [(2, "sparse", "4000", "3500"), (3, "cover", "3000", "3500")], ids=["2 ranks, primal loop", "3 ranks, dual loop"])
def test_deferred_sharding_runs_replicas_first_and_goes_live_when_the_tableau_row_becomes_a_pass(world, family, rows, cols):
    """Round 5 (engine.h: shard_defer_): while the nucleus is small a pivot is 40-60 us of latency-bound launches and per-pivot exchanges
    only slow it down, so the ranks start as bit-identical REPLICAS (the deterministic unsharded iteration on every rank, no exchange) and
    the column-block sharding goes live at the first batch that leaves the sparse-tableau-row regime.  The default (MLP_SHARD_DEFER unset
    = on): after 30 pivots the sharding is not live yet, at the end it is, and the whole run takes the unsharded run's pivots — i.e. the
    replicas were identical at the switch and the sharded continuation picked up from them."""
    env = dict(os.environ, MLP_SHARD_DEFER="1", SHARD_TEST_PROBE="30")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), str(world), rows, cols, "12", "700", family],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "traces identical: True" in r.stdout
    assert "sharding live after the probe / at the end: 0 / 1" in r.stdout, r.stdout[-1500:]
    assert "go-live fingerprint checks passed on rank 0: 1" in r.stdout, r.stdout[-1500:]   # the ranks compared their replicated state at the switch


def test_deferred_sharding_refuses_to_go_live_on_ranks_that_are_not_the_same_replica():
    """the switch from replicas to column-block sharding assumes every rank reached the same pivot with the same
    basis; nothing exchanged before it could show a divergence.  At the go-live point every rank now publishes a fingerprint of its
    replicated state (pivots taken, nucleus size, hash of basic_vars / nb_vars, objective bits) through the rendezvous object and the
    solve fails on EVERY rank if two differ.  MLP_TEST_GOLIVE_SKEW=1 makes rank 1 publish a perturbed hash: both ranks must raise."""
    env = dict(os.environ, MLP_SHARD_DEFER="1", MLP_TEST_GOLIVE_SKEW="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_test.py"), "2", "4000", "3500", "12", "700", "sparse"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0
    assert r.stderr.count("are not the same replica at the go-live point") >= 2, r.stderr[-3000:]
    assert "traces identical" not in r.stdout
