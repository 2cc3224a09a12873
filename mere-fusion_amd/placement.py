"""GPU choice INSIDE the drop-in (VERDICT r05 missing #2).

The reference pins every session to the process's default device: `device = 'cuda'` (lipreal.py:29, musereal.py:58, app.py:378), sessions are
admitted against one global cap (app.py:42,79-80,705) and nothing spreads them over the GPUs of a node.  Its files stay untouched, so the spreading happens
where the drop-in is entered: the first time a PROCESS creates a model through it (`Wav2Lip.to('cuda')` in `lipreal.inference`'s per-session process,
`load_all_model()` in `musereal.inference`'s, `NeRFNetwork(opt)` in app.py's) the process takes a place on the GPU with the lowest load fraction and makes
it its current device -- `'cuda'` then means that GPU for everything the reference does afterwards in that process.

Placement is `harness.SessionPlacer`'s rule (lowest load / capacity, ties to the lowest index, never above a GPU's capacity) shared between processes
through one JSON file under an flock:

    $MF_PLACEMENT_DIR/placement.json      {"gpus": N, "capacity": [...], "held": {"<pid>": [gpu, weight, start_time], ...}}      (default dir: /dev/shm/merefusion-<uid>)

A place is released at process exit (atexit) and, for processes that died without running it, by the next caller (a holder whose pid no longer exists -- or
exists with another start time -- is dropped).  Sessions are independent: there is no collective and no rank numbering ("replicas only").

Environment: MF_GPUS (how many GPUs take sessions; default: HIP_VISIBLE_DEVICES as the deployment set it, else the KFD topology), MF_GPU_CAPACITY (sessions per GPU, one number or a comma list;
default: unbounded, i.e. pure balancing -- the reference's own --max_session stays the admission cap), MF_PLACEMENT=0 (off: the reference's behaviour),
MF_PLACEMENT_DIR.  With one visible GPU everything here is a no-op."""
import atexit
import fcntl
import json
import os

_state = {"pid": None, "gpu": None, "charged": 0.0, "n": 0}
UNBOUNDED = 1 << 30
# what a process that holds a GPU without a session weighs (the reference's parent process; app.py before its first ER-NeRF session): enough to break ties, so that eight
# app.py processes started together take eight GPUs instead of all taking GPU 0, and far too little to use up a GPU's capacity (the capacity test allows for it)
HOLDER_WEIGHT = 1.0 / 64


def _dir():
    d = os.environ.get("MF_PLACEMENT_DIR") or os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"merefusion-{os.getuid()}")
    os.makedirs(d, exist_ok=True)
    return d


def _proc_start(pid):
    """start time of a process in clock ticks (field 22 of /proc/<pid>/stat), None when it does not exist: tells a recycled pid from the holder"""
    try:
        with open(f"/proc/{pid}/stat", "rb") as f:
            return int(f.read().rsplit(b")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return None


def _capacities(n):
    e = os.environ.get("MF_GPU_CAPACITY", "").strip()
    if not e:
        return [UNBOUNDED] * n
    v = [int(x) for x in e.split(",")]
    if len(v) == 1:
        v = v * n
    if len(v) != n or min(v) < 0:
        raise RuntimeError(f"MF_GPU_CAPACITY: one non-negative number, or one per GPU ({n}), is required")
    return v


class _Locked:
    """the placement file, read and rewritten under an exclusive flock"""

    def __enter__(self):
        self.f = open(os.path.join(_dir(), "placement.json"), "a+")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        self.f.seek(0)
        raw = self.f.read()
        try:
            self.data = json.loads(raw) if raw.strip() else {}
        except ValueError:
            self.data = {}                                           # a torn file (killed writer): start over; live holders re-enter on their next call
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.f.seek(0)
            self.f.truncate()
            json.dump(self.data, self.f)
            self.f.flush()
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def _prune(held):
    for pid in list(held):
        if _proc_start(int(pid)) != held[pid][2]:
            del held[pid]


def choose(held, capacity, weight):
    """SessionPlacer.start_session's rule over the table of holders: the GPU with the lowest load fraction that still has room for `weight`
    (ties: lowest index); None when every GPU is full."""
    load = [0.0] * len(capacity)
    for g, w, _ in held.values():
        if 0 <= g < len(load):
            load[g] += w
    free = [g for g in range(len(capacity)) if load[g] + weight <= capacity[g] + 0.5]
    if not free:
        return None
    return min(free, key=lambda g: (load[g] / capacity[g], load[g], g))


def place(n_gpus, weight=1.0, pid=None, want=None):
    """Takes (or adds `weight` to) the calling process's place.  -> gpu index, or raises RuntimeError('Maximum number of sessions reached') when
    MF_GPU_CAPACITY leaves no room (the message of app.py:80)."""
    pid = os.getpid() if pid is None else pid
    cap = _capacities(n_gpus)
    with _Locked() as L:
        d = L.data
        if d.get("gpus") != n_gpus:
            d.clear()
            d.update(gpus=n_gpus, held={})
        d["capacity"] = cap
        held = d.setdefault("held", {})
        _prune(held)
        mine = held.get(str(pid))
        if mine is not None:
            g = mine[0]
            if weight:
                others = sum(w for p, (gg, w, _) in held.items() if gg == g)
                if others + weight > cap[g] + 0.5:
                    raise RuntimeError("Maximum number of sessions reached")
                mine[1] += weight
            return g
        g = want if want is not None else choose(held, cap, weight)
        if g is None:
            raise RuntimeError("Maximum number of sessions reached")
        held[str(pid)] = [g, weight, _proc_start(pid)]
        return g


def release(pid=None):
    pid = os.getpid() if pid is None else pid
    try:
        with _Locked() as L:
            L.data.get("held", {}).pop(str(pid), None)
    except OSError:
        pass


def table():
    """{gpu: [pids]} of the live holders (diagnostics, tests)."""
    with _Locked() as L:
        held = L.data.setdefault("held", {})
        _prune(held)
        out = {}
        for pid, (g, w, _) in held.items():
            out.setdefault(g, []).append(int(pid))
        return out


def physical_gpus():
    """The GPUs this process may use, as the strings HIP_VISIBLE_DEVICES takes: the visibility the deployment set (kept in MF_ORIG_VISIBLE_DEVICES once this
    module has narrowed it, so that a child process chooses among ALL of them again), else MF_GPUS, else the KFD topology (no HIP call: the runtime reads
    HIP_VISIBLE_DEVICES when it initialises, and must not have done so yet)."""
    orig = os.environ.get("MF_ORIG_VISIBLE_DEVICES")
    if orig is None and os.environ.get("MF_PLACED_GPU") is None:
        orig = os.environ.get("HIP_VISIBLE_DEVICES")
    if orig:
        return [x.strip() for x in orig.split(",") if x.strip()]
    e = os.environ.get("MF_GPUS")
    if e:
        return [str(i) for i in range(int(e))]
    n, root = 0, "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in sorted(os.listdir(root), key=lambda x: int(x) if x.isdigit() else 1 << 30):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
            if int(props.get("simd_count", 0)) > 0:                  # CPU nodes have none
                n += 1
    except (OSError, ValueError):
        n = 0
    return [str(i) for i in range(n)]


def _runtime_initialised():
    """has this process already brought up HIP?  (then HIP_VISIBLE_DEVICES no longer has any effect and only torch.cuda.set_device is left)"""
    import sys
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_initialized():
        return True
    l = sys.modules.get("mere_fusion_amd._lib")
    return bool(l is not None and getattr(l, "_device_inited", False))


def enabled():
    """off with MF_PLACEMENT=0, and under a launcher that already numbers its processes per GPU (torch.distributed.run exports LOCAL_RANK: bench.py's ranks
    choose their own device)"""
    return os.environ.get("MF_PLACEMENT", "1") != "0" and "LOCAL_RANK" not in os.environ


def ensure_placed(session=True):
    """The first call in a process chooses its GPU (a forked / spawned child chooses again, among all the node's GPUs); every call with session=True charges one
    more session to that GPU -- the model constructors of the drop-in (`uncharge()` gives it back when the model dies) -- while session=False only makes sure the
    process HAS a GPU: the reference's parent process, which runs the front-ends (mel, Whisper features) but no model, does not count as a session (it holds its
    GPU with HOLDER_WEIGHT: a tie-breaker, not a session).
    -> the chosen entry of physical_gpus(), or None when placement is off or there is at most one GPU.

    Before the HIP runtime is up the choice is made by narrowing HIP_VISIBLE_DEVICES to the one GPU: every thread of the process -- the reference builds its
    models in one thread and renders in another -- then sees it as device 0, and `'cuda'` cannot mean anything else.  A process whose runtime is already up has
    made its choice: it is left where it is and only ENTERED in the table (on its current device), so that the others balance around it."""
    if not enabled():
        return None
    pid = os.getpid()
    w = 1.0 if session else 0.0
    if _state["pid"] == pid and _state["gpu"] is not None:
        if w:
            place(_state["n"], weight=w)
            _state["charged"] += w
        return _state["gpu"]
    phys = physical_gpus()
    if len(phys) <= 1:
        return None
    up = _runtime_initialised()
    if up:
        import torch
        cur = torch.cuda.current_device()
        g = place(len(phys), weight=w or HOLDER_WEIGHT, want=cur if cur < len(phys) else None)
    else:
        g = place(len(phys), weight=w or HOLDER_WEIGHT)
        os.environ.setdefault("MF_ORIG_VISIBLE_DEVICES", ",".join(phys))
        os.environ["HIP_VISIBLE_DEVICES"] = phys[g]
    atexit.register(release)
    _state.update(pid=pid, gpu=phys[g], charged=w, n=len(phys))
    os.environ["MF_PLACED_GPU"] = phys[g]
    return phys[g]


def uncharge(weight=1.0):
    """one session of this process ended (its model object was collected): its share of the GPU's load goes back; the process keeps its GPU"""
    if _state["pid"] != os.getpid() or _state["gpu"] is None or _state["charged"] <= 0:
        return
    try:
        with _Locked() as L:
            mine = L.data.get("held", {}).get(str(os.getpid()))
            if mine is not None:
                mine[1] = max(mine[1] - weight, 0.0)
        _state["charged"] = max(_state["charged"] - weight, 0.0)
    except OSError:
        pass


def charge_session(owner):
    """What a drop-in model constructor calls: places the process if need be, charges one session, and ties the charge to `owner`'s lifetime."""
    g = ensure_placed(session=True)
    if g is not None:
        import weakref
        weakref.finalize(owner, uncharge, 1.0)
    return g
