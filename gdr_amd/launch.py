"""One command fans out over the GPUs of a node: start N fresh rank processes (one per GPU) of a script or module.

The reference's analogue is Data_process/NQ_dataset/bert/bert_NQ.sh:5-12 (one `python bert.py --idx i` per GPU, launched
by one shell script) and Lightning's DDP spawn behind `main.py --n_gpu N` (main.py:57-70).  Here the parent starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` as a CHILD
process and relays its output: the parent itself must not have initialised the GPU (importing torch is fine) and never
replaces itself with another program — on this pool an exec from a process that holds the GPU takes the machine down.
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def under_launcher():
    """True inside a rank started by torch.distributed.run (it exports RANK / WORLD_SIZE / MASTER_PORT)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _stat(pid):
    """(parent pid, start time in clock ticks) of a live process from /proc/<pid>/stat (fields 4 and 22), else None."""
    try:
        with open(f"/proc/{pid}/stat") as f:
            rest = f.read().rsplit(")", 1)[1].split()
        return int(rest[1]), int(rest[19])
    except (OSError, ValueError, IndexError):
        return None


def descendants(pid, with_start=False):
    """PIDs of every live descendant of `pid` (children, grandchildren, ...) from /proc/<pid>/stat's parent field; with_start:
    {pid: start time} — a PID alone does not name a process (it is recycled), (pid, start time) does."""
    kids, started = {}, {}
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        st = _stat(name)
        if st is None:
            continue
        kids.setdefault(st[0], []).append(int(name))
        started[int(name)] = st[1]
    out, todo = [], [pid]
    while todo:
        for c in kids.get(todo.pop(), []):
            out.append(c)
            todo.append(c)
    return {c: started[c] for c in out} if with_start else out


def kill_if_same(pid, start, sig=signal.SIGKILL):
    """Signals `pid` only if it is still the process that was collected (same start time): a PID that exited and was handed to an
    unrelated process during the grace period is left alone.  Returns True if a signal was sent."""
    st = _stat(pid)
    if st is None or st[1] != start:
        return False
    try:
        os.kill(pid, sig)
        return True
    except (ProcessLookupError, PermissionError):
        return False


def spawn_ranks(n, argv, script=None, module=None, env=None, relay=True, timeout=None):
    """Runs `script argv...` (or `-m module argv...`) as n ranks under torch.distributed.run, as a child process.
    Returns (returncode, stdout text).  relay=True echoes every stdout line of the ranks to this process' stdout as it
    arrives; relay=callable hands each line to the callable instead (bench.py lets only rank 0's JSON line through to
    stdout and sends the rest — RCCL's version banner, progress prints — to stderr); stderr is inherited."""
    if (script is None) == (module is None):
        raise ValueError("spawn_ranks: give exactly one of script= / module=")
    if n < 1:
        raise ValueError("spawn_ranks: n must be >= 1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())]
    cmd += ["-m", module] if module is not None else [script]
    cmd += list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
    e.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        e.pop(k, None)                                   # the children get their own
    # torch.distributed.run puts every rank into a session of its own, so neither a SIGKILL to the launcher nor one to its process
    # group reaches them: the ranks would live on as orphans that hold the GPUs AND the inherited stdout pipe, and the read loop
    # below would never end.  stop() therefore walks the launcher's process tree and kills exactly those PIDs.
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=e, text=True, bufsize=1, start_new_session=True)
    lines = []

    def stop(grace=5.0):
        """SIGTERM to the launcher (torch.distributed.run forwards it to its workers), then SIGKILL to (1) the launcher's own process
        group — it leads a session of its own (start_new_session), so the group holds the launcher and whatever it started without
        a new session — and (2) every process of its tree, each identified by (pid, start time) and collected before and during the
        grace period (a rank re-parented to init is no longer findable through its parent; a PID recycled meanwhile is not touched)."""
        tree = descendants(proc.pid, with_start=True)
        try:
            proc.terminate()
            t_end = time.monotonic() + grace
            while proc.poll() is None and time.monotonic() < t_end:
                time.sleep(0.05)
                for pid, st in descendants(proc.pid, with_start=True).items():
                    tree.setdefault(pid, st)
        finally:
            if proc.poll() is None:                      # still ours: its pgid == its pid (session leader), not a recycled one
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
            for pid, st in sorted(tree.items()):
                kill_if_same(pid, st)

    # the read loop below only ends when the ranks close their stdout: a watchdog enforces `timeout` on a hung launch
    watchdog = threading.Timer(timeout, stop) if timeout else None
    if watchdog:
        watchdog.daemon = True
        watchdog.start()
    try:
        for line in proc.stdout:
            lines.append(line)
            if callable(relay):
                relay(line)
            elif relay:
                sys.stdout.write(line)
                sys.stdout.flush()
        rc = proc.wait()
    except BaseException:
        stop()
        proc.wait()
        raise
    finally:
        if watchdog:
            watchdog.cancel()
    return rc, "".join(lines)
