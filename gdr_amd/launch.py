"""One command fans out over the GPUs of a node: start N fresh rank processes (one per GPU) of a script or module.

The reference's analogue is Data_process/NQ_dataset/bert/bert_NQ.sh:5-12 (one `python bert.py --idx i` per GPU, launched
by one shell script) and Lightning's DDP spawn behind `main.py --n_gpu N` (main.py:57-70).  Here the parent starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` as a CHILD
process and relays its output: the parent itself must not have initialised the GPU (importing torch is fine) and never
replaces itself with another program — on this pool an exec from a process that holds the GPU takes the machine down.
"""
import os
import socket
import subprocess
import sys
import threading


def under_launcher():
    """True inside a rank started by torch.distributed.run (it exports RANK / WORLD_SIZE / MASTER_PORT)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


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
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=e, text=True, bufsize=1)
    lines = []
    # the read loop below only ends when the ranks close their stdout: a watchdog enforces `timeout` on a hung launch
    watchdog = threading.Timer(timeout, proc.kill) if timeout else None
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
        proc.kill()
        proc.wait()
        raise
    finally:
        if watchdog:
            watchdog.cancel()
    return rc, "".join(lines)
