"""One-process-per-GPU helpers for the sharded pricing path.

Control plane only (no arithmetic).  The pivot kernels exchange their per-pivot candidates through mailboxes
that live in each GPU's own HBM and are written by the peers over xGMI (HIP IPC peer mappings); the IPC handles
travel through a small POSIX shared-memory rendezvous object created and zeroed by rank 0, whose first part
doubles as the host-memory mailbox of the fallback transport (MLP_TRANSPORT=host).  `torch.distributed` (RCCL on
GPUs, gloo in the CPU tests) carries the object's name, the barriers and the timing reductions.
"""
import os
import uuid

MAILREC_BYTES = 64
KINDS, PARITIES = 6, 2  # kernels.h: MAIL_KINDS (pricing, primal ratio decision, dual ratio min, dual ratio candidate, W-stream flags, handshake)


RENDEZVOUS_BYTES = 128  # engine.hip: one record per rank (ready flag, device, pid, 64-byte HIP IPC handle)


def host_box_bytes(world):
    return MAILREC_BYTES * KINDS * PARITIES * world


def mailbox_bytes(world):
    """Size of the shared-memory object: host-transport mailbox + the rendezvous records (896 bytes per rank)."""
    return host_box_bytes(world) + RENDEZVOUS_BYTES * world


def shard_range(n, rank, world):
    """Non-basic positions [lo, hi) owned by `rank` — must match Engine::sync_view."""
    return (n * rank) // world, (n * (rank + 1)) // world


def create_mailbox(world, name=None):
    """Rank 0: create a zero-filled shared-memory object; returns its shm name ('/mlp_<uuid>')."""
    name = name or "/mlp_" + uuid.uuid4().hex[:16]
    path = "/dev/shm" + name
    with open(path, "wb") as f:
        f.write(b"\0" * mailbox_bytes(world))
    return name


def remove_mailbox(name):
    try:
        os.unlink("/dev/shm" + name)
    except OSError:
        pass


def _try_enable(solution, dist, rank, world, transport=None):
    """One collective attempt: rank 0 creates the rendezvous object, every rank enables sharding; returns
    (box name, list of per-rank error strings).  transport None: the engine's default (device mailboxes written by the peers,
    or what MLP_TRANSPORT says); "rccl": rank 0 also makes the ncclUniqueId that travels with the box name."""
    uid = None
    if transport == "rccl" and rank == 0:
        import minilp_amd as _M
        try:
            uid = _M.api.rccl_unique_id()
        except Exception as e:
            uid = f"error: {e}"
    box = [create_mailbox(world) if rank == 0 else None, uid]
    dist.broadcast_object_list(box, src=0)
    err = None
    try:
        if transport is None:
            solution.enable_sharding(rank, world, box[0])
        elif transport == "rccl" and not isinstance(box[1], (bytes, bytearray)):
            raise RuntimeError(f"rank 0 could not make an ncclUniqueId ({box[1]})")
        else:
            solution.enable_sharding_ex(rank, world, box[0], transport, box[1] if transport == "rccl" else None)
    except Exception as e:  # every rank must leave through the same door
        err = f"rank {rank}: {e}"
    errs = [None] * world
    dist.all_gather_object(errs, err)  # doubles as the barrier after which every box is mapped
    return box[0], [e for e in errs if e]


def setup_sharding(solution, dist=None):
    """Call on every rank after `Problem.solve(budget=0)`: agree on a rendezvous object and enable sharding.
    `dist` is torch.distributed (initialised) or None for a single process.  Returns the object's name (rank 0
    should remove it at the end).

    Transport: device mailboxes written by the peers over xGMI (HIP IPC).  If ANY rank cannot set that up (IPC or peer
    access unavailable on the node, or the handshake does not deliver), all ranks together fall back to the RCCL transport
    (records delivered by ncclAllGather, pumped by the host while a batch is in flight) and, failing that too, to the host-memory mailbox — still the same
    sharded solve of one LP, only the 64-byte exchanges take the PCIe route; `Solution.transport()` says which one is
    in use.  If that fails too the call raises on every rank (never a silent change of what is being run)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    forced = os.environ.get("MLP_TRANSPORT")  # "peer" / "host" / "rccl" / "pump": no fallback chain, that transport or an error
    name, bad = _try_enable(solution, dist, rank, world, forced if forced in ("rccl", "pump", "host") else None)
    if bad and not forced:
        # fallback chain: peer stores -> RCCL all-gather pump (ranks on distinct devices) -> host-memory mailbox
        if rank == 0:
            remove_mailbox(name)
            print("[minilp_amd.dist] peer transport unavailable (" + "; ".join(bad) + "): trying the RCCL all-gather transport", flush=True)
        name, bad = _try_enable(solution, dist, rank, world, "rccl")
        if bad:
            if rank == 0:
                remove_mailbox(name)
                print("[minilp_amd.dist] RCCL transport unavailable (" + "; ".join(bad) + "): falling back to the host-memory mailbox", flush=True)
            name, bad = _try_enable(solution, dist, rank, world, "host")
    if bad:
        if rank == 0:
            remove_mailbox(name)
        raise RuntimeError("sharding could not be enabled on every rank: " + "; ".join(bad))
    return name


def combine_candidates(cands):
    """Reference implementation of the on-device candidate reduction (score desc, position asc;
    position -1 = no candidate) — used by the CPU tests to pin the tie-break rule."""
    best = (-float("inf"), -1)
    for score, pos in cands:
        if pos < 0:
            continue
        if best[1] < 0 or score > best[0] or (score == best[0] and pos < best[1]):
            best = (score, pos)
    return best


def transport_name(solution=None):
    """Human-readable name of the per-pivot exchange transport (bench.py's config.parallelism)."""
    return solution.transport() if solution is not None else "none"
