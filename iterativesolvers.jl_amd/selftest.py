"""Transport self-test of the row-partitioned path -- one CHILD PROCESS per rank and transport, started by ``bench.py --gpus N``
(``bench_dist.transport_selftest``) BEFORE anything is timed, and usable by any host (``python selftest.py --help``).

Why a child: the first contact of RCCL / HIP IPC with a new machine can hang inside a library call (communicator bootstrap, a
collective whose peer never arrives), and a thread stuck in a C call cannot be cancelled.  A child can be killed: a transport whose
self-test fails or does not return is dropped from the candidates and the parent never touches it.

Per transport, every check bit-exact and timed:
  mailbox   (include/mik.h "Transport 3": no RCCL)
     mailbox_scalars   64 sequence-numbered rounds of one value per rank through the peer-mapped mailbox slots (mik_comm_allgather_sum on a
                       communicator without RCCL: store into every peer's slot, rank-ordered sum) -- the two scalar sums of a cg! step
     landing_4MB       6 rounds of a 4 MB payload (> one XCD's L2) per neighbour through the library's landing buffers (mik_plink_exchange:
                       push kernel over the link, flag, landing copy with system-scope loads); every 64-bit word carries sender, receiver,
                       round and index and is compared on arrival, per neighbour pair
  rccl      (include/mik.h "Transport 1")
     rccl_allgather    64 rounds of ncclAllGather of one double per rank (mik_comm_allgather_sum)
     rccl_halo_4MB     6 rounds of ncclSend / ncclRecv of the same 4 MB payloads (mik_comm_halo), compared per neighbour pair

The ranks of one self-test meet through files in a directory the parent names (rank 0's ncclUniqueId, the 64-byte IPC handles): the
test does not depend on the parent's own bootstrap channel.  Every wait is bounded; the result is ONE JSON line on stdout.
The child loads PyTorch first when it is there (--no-torch: not), so that it runs on the HIP runtime and RCCL the parent uses."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


class Meet:
    """file rendezvous of the ranks of one self-test (bounded waits)"""

    def __init__(self, directory, rank, world, deadline):
        self.dir, self.rank, self.world, self.deadline = directory, rank, world, deadline
        os.makedirs(directory, exist_ok=True)

    def put(self, name, payload: bytes):
        tmp = os.path.join(self.dir, f".{name}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, os.path.join(self.dir, name))

    def get(self, name) -> bytes:
        path = os.path.join(self.dir, name)
        while not os.path.exists(path):
            if time.monotonic() > self.deadline:
                raise TimeoutError(f"rank {self.rank}: timed out waiting for {name}")
            time.sleep(0.002)
        with open(path, "rb") as f:
            return f.read()

    def gather(self, tag, payload: bytes):
        self.put(f"{tag}.{self.rank}", payload)
        return [self.get(f"{tag}.{q}") for q in range(self.world)]

    def barrier(self, tag):
        self.gather(tag, b"1")


def ring_plan(rank, world, M):
    """z-slab neighbours: previous and next rank, M entries each way.  Returns (recv, send, dst): recv / send segments (peer, offset, count);
    dst[i] = where send segment i lands in the receiver's ghost region."""
    peers = [q for q in (rank - 1, rank + 1) if 0 <= q < world]
    recv = [(q, i * M, M) for i, q in enumerate(peers)]
    send = [(q, i * M, M) for i, q in enumerate(peers)]
    dst = []
    for q in peers:
        theirs = [p for p in (q - 1, q + 1) if 0 <= p < world]
        dst.append(theirs.index(rank) * M)
    return recv, send, dst


def pattern(sender, receiver, rnd, M):
    """64-bit words that name sender, receiver, round and index (small normal doubles when viewed as fp64)"""
    head = np.uint64(((sender + 1) << 56) | ((receiver + 1) << 48) | ((rnd & 0xFFFF) << 32))
    return head | np.arange(M, dtype=np.uint64)


def run(args):
    t_start = time.monotonic()
    deadline = t_start + args.timeout
    sys.path.insert(0, ROOT)
    runtime = "system (libmik.so's RUNPATH)"
    if not args.no_torch:
        # the parent (bench.py) runs inside PyTorch-ROCm, whose bundled HIP runtime and RCCL libmik.so then shares; the child must prove the
        # transports on the SAME libraries, so it loads them first (costs about two seconds)
        try:
            import torch  # noqa: F401
            runtime = f"the one PyTorch {torch.__version__} bundles"
        except Exception as exc:      # noqa: BLE001
            runtime += f" (import torch failed: {type(exc).__name__})"
    import __graft_entry__ as graft
    pkg = graft.load_package()
    L = pkg.lib()
    check = pkg._lib.check
    rank, world, M = args.rank, args.world, args.bytes // 8
    meet = Meet(args.dir, rank, world, deadline)
    out = {"transport": args.transport, "rank": rank, "world": world, "device": args.device, "checks": {}, "pass": False, "failure": None}
    os.environ.setdefault("MIK_MAILBOX_TIMEOUT_MS", str(int(min(20.0, args.timeout / 3) * 1000)))
    ctx = pkg.HipContext(args.device)
    out["machine"] = {k: v for k, v in ctx.info().items() if k in ("compute_units", "xcds", "arch")}
    out["hip_runtime_and_rccl"] = runtime
    ident = None
    if args.transport == "rccl":
        if rank == 0:
            buf = C.create_string_buffer(128)
            check(L.mik_comm_unique_id(buf), "mik_comm_unique_id")
            meet.put("nccl_id", bytes(buf.raw))
        ident = meet.get("nccl_id")
    comm = C.c_void_p()
    t0 = time.perf_counter()
    check(L.mik_comm_create(ctx.handle, ident, rank, world, C.byref(comm)), "mik_comm_create", ctx.handle)
    out["comm_create_seconds"] = time.perf_counter() - t0
    lp, ip = C.POINTER(C.c_int64), C.POINTER(C.c_int)
    recv, send, dst = ring_plan(rank, world, M)
    link = C.c_void_p()
    if args.transport == "mailbox":
        mine = C.create_string_buffer(64)
        check(L.mik_comm_mailbox_export(comm, mine), "mik_comm_mailbox_export", ctx.handle)
        handles = b"".join(meet.gather("mailbox", bytes(mine.raw)))
        check(L.mik_comm_mailbox_connect(comm, handles), "mik_comm_mailbox_connect", ctx.handle)
        meet.barrier("mailbox_connected")
        a, b = C.c_int(), C.c_int()
        L.mik_comm_mailbox_info(comm, C.byref(a), C.byref(b))
        out["mailbox"] = {"connected": bool(a.value), "finegrained_memory": bool(b.value)}

    # ---- one value per rank, sequence-numbered, summed in rank order -------------------------------------------------------------------
    name = "mailbox_scalars" if args.transport == "mailbox" else "rccl_allgather"
    ok, worst = True, None
    v = np.zeros(2, np.float64)
    lat = []
    for r in range(args.rounds):
        v[0], v[1] = (rank + 1) * 1024.0 + r + 0.25, -(r + 1.0) * (rank + 2)
        t0 = time.perf_counter()
        check(L.mik_comm_allgather_sum(comm, 0, 2, v.ctypes.data_as(C.c_void_p)), "mik_comm_allgather_sum", ctx.handle)
        lat.append(time.perf_counter() - t0)
        want0 = want1 = 0.0
        for q in range(world):                                          # ((p_0 + p_1) + p_2) + ...
            a0, a1 = (q + 1) * 1024.0 + r + 0.25, -(r + 1.0) * (q + 2)
            want0, want1 = (a0, a1) if q == 0 else (want0 + a0, want1 + a1)
        if v[0] != want0 or v[1] != want1:
            ok, worst = False, f"round {r}: got {v.tolist()}, expected {[want0, want1]}"
            break
    out["checks"][name] = {"pass": ok, "rounds": len(lat), "us_per_round_median": float(np.median(lat) * 1e6), "us_first_round": float(lat[0] * 1e6),
                           "what": "host -> device -> every peer -> rank-ordered sum -> host, one call per round", **({"failure": worst} if worst else {})}

    # ---- 4 MB per neighbour, every word checked --------------------------------------------------------------------------------------
    name = "landing_4MB" if args.transport == "mailbox" else "rccl_halo_4MB"
    rec = {"pass": True, "bytes_per_neighbour": M * 8, "per_neighbour": {}}
    if recv:
        rp = np.asarray([s[0] for s in recv], np.int32); ro = np.asarray([s[1] for s in recv], np.int64); rc = np.asarray([s[2] for s in recv], np.int64)
        sp = np.asarray([s[0] for s in send], np.int32); so = np.asarray([s[1] for s in send], np.int64); sc = np.asarray([s[2] for s in send], np.int64)
        n_ghost = int(M * len(recv))
        if args.transport == "mailbox":
            check(L.mik_plink_create(comm, 0, n_ghost, rp.size, rp.ctypes.data_as(ip), ro.ctypes.data_as(lp), rc.ctypes.data_as(lp),
                                     sp.size, sp.ctypes.data_as(ip), so.ctypes.data_as(lp), sc.ctypes.data_as(lp), C.byref(link)), "mik_plink_create", ctx.handle)
            hb = C.create_string_buffer(64)
            check(L.mik_plink_export(link, hb), "mik_plink_export", ctx.handle)
        else:
            hb = C.create_string_buffer(64)
        info = [json.loads(x) for x in meet.gather("link", json.dumps({"h": bytes(hb.raw).hex(), "g": n_ghost if recv else 0}).encode())]
        if args.transport == "mailbox":
            handles = b"".join(bytes.fromhex(i["h"]) for i in info)
            counts = np.asarray([i["g"] for i in info], np.int64)
            d = np.asarray(dst, np.int64)
            check(L.mik_plink_connect(link, handles, counts.ctypes.data_as(lp), d.ctypes.data_as(lp)), "mik_plink_connect", ctx.handle)
            meet.barrier("link_connected")
            a, b, g = C.c_int(), C.c_int(), C.c_int64()
            L.mik_plink_info(link, C.byref(a), C.byref(b), C.byref(g))
            rec["landing_buffer"] = {"connected": bool(a.value), "finegrained_memory": bool(b.value)}
        sendv = pkg.HipVector(M * len(send), np.float64, ctx)
        ghost = pkg.HipVector(n_ghost, np.float64, ctx)
        lat = []
        for r in range(args.payload_rounds):
            host = np.concatenate([pattern(rank, q, r, M) for q, _, _ in send])
            sendv.copy_from_host(host.view(np.float64))
            ghost.fill_(0)
            ctx.synchronize()
            meet.barrier(f"round{r}")                                   # (timing only: every rank enters the exchange together)
            t0 = time.perf_counter()
            if args.transport == "mailbox":
                check(L.mik_plink_exchange(link, C.c_void_p(sendv.ptr), C.c_void_p(ghost.ptr)), "mik_plink_exchange", ctx.handle)
            else:
                check(L.mik_comm_halo(comm, 0, C.c_void_p(sendv.ptr), C.c_void_p(ghost.ptr), rp.size, rp.ctypes.data_as(ip), ro.ctypes.data_as(lp),
                                      rc.ctypes.data_as(lp), sp.size, sp.ctypes.data_as(ip), so.ctypes.data_as(lp), sc.ctypes.data_as(lp)), "mik_comm_halo", ctx.handle)
                ctx.synchronize()
            lat.append(time.perf_counter() - t0)
            got = ghost.to_numpy().view(np.uint64)
            for (q, off, cnt) in recv:
                want = pattern(q, rank, r, M)
                bad = int(np.count_nonzero(got[off:off + cnt] != want))
                pr = rec["per_neighbour"].setdefault(f"{q}->{rank}", {"pass": True, "rounds": 0, "words_checked": 0, "words_wrong": 0})
                pr["rounds"] += 1
                pr["words_checked"] += int(cnt)
                pr["words_wrong"] += bad
                if bad:
                    pr["pass"] = rec["pass"] = False
                    first = int(np.flatnonzero(got[off:off + cnt] != want)[0])
                    pr.setdefault("first_wrong", {"round": r, "index": first, "got": hex(int(got[off + first])), "expected": hex(int(want[first]))})
        # latency: back-to-back exchanges of the last payload without the file barrier in between (a symmetric plan needs none: a rank cannot run
        # two exchanges ahead of its neighbours), one blocking call each
        meet.barrier("timing")
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            if args.transport == "mailbox":
                check(L.mik_plink_exchange(link, C.c_void_p(sendv.ptr), C.c_void_p(ghost.ptr)), "mik_plink_exchange", ctx.handle)
            else:
                check(L.mik_comm_halo(comm, 0, C.c_void_p(sendv.ptr), C.c_void_p(ghost.ptr), rp.size, rp.ctypes.data_as(ip), ro.ctypes.data_as(lp),
                                      rc.ctypes.data_as(lp), sp.size, sp.ctypes.data_as(ip), so.ctypes.data_as(lp), sc.ctypes.data_as(lp)), "mik_comm_halo", ctx.handle)
                ctx.synchronize()
        per = (time.perf_counter() - t0) / reps
        got = ghost.to_numpy().view(np.uint64)          # ... and the last one still carries the right words
        for (q, off, cnt) in recv:
            if np.count_nonzero(got[off:off + cnt] != pattern(q, rank, args.payload_rounds - 1, M)):
                rec["pass"] = rec["per_neighbour"][f"{q}->{rank}"]["pass"] = False
                rec["per_neighbour"][f"{q}->{rank}"]["failure"] = "wrong words after the back-to-back exchanges"
        rec.update(rounds=len(lat), us_first_exchange=float(lat[0] * 1e6), us_per_exchange_median=float(per * 1e6), back_to_back_exchanges_timed=reps,
                   gbs_received=float(M * 8 * len(recv) / per / 1e9),
                   what="blocking call: push kernel into the neighbours' landing buffers + flag + landing copy + host wait" if args.transport == "mailbox"
                        else "ncclSend / ncclRecv group on the ctx stream + host wait")
    else:
        rec["note"] = "a world of one has no neighbour"
    out["checks"][name] = rec
    meet.barrier("done")
    if link:
        L.mik_plink_destroy(link)
    L.mik_comm_destroy(comm)
    out["pass"] = all(c["pass"] for c in out["checks"].values())
    out["seconds"] = time.monotonic() - t_start
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--transport", choices=["mailbox", "rccl"], required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--device", type=int, required=True)
    ap.add_argument("--dir", required=True, help="directory shared by the ranks of this self-test (created if absent)")
    ap.add_argument("--rounds", type=int, default=64)
    ap.add_argument("--payload-rounds", type=int, default=6)
    ap.add_argument("--bytes", type=int, default=4 << 20, help="payload per neighbour (default 4 MB: larger than one XCD's L2)")
    ap.add_argument("--timeout", type=float, default=60.0)
    ap.add_argument("--no-torch", action="store_true", help="do not load PyTorch's HIP runtime / RCCL first (a host without PyTorch)")
    args = ap.parse_args()
    try:
        out = run(args)
    except BaseException as exc:      # noqa: BLE001 -- the parent wants a line whatever happened
        out = {"transport": args.transport, "rank": args.rank, "pass": False, "failure": f"{type(exc).__name__}: {exc}"[:400]}
    print(json.dumps(out), flush=True)
    sys.exit(0 if out.get("pass") else 1)


if __name__ == "__main__":
    main()
