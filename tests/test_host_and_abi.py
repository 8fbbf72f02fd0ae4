"""CPU-side tests: the C-ABI library loads and exports every symbol the header declares (no compute calls),
host logic against the reference's golden vectors, config parsing, and the multi-process gather path (2 and 8 ranks)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from metdetpy_amd import _lib, utils as U
from metdetpy_amd.metstruct import BinaryCfg, CollectorCfg, DetectorCfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "metdet_hip.h")).read()
    declared = set(re.findall(r"\b(mdh_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"mdh_ctx", "mdh_cfg", "mdh_frame_info"}
    assert declared, "no declarations parsed"
    L = _lib.load()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(_lib.EXPORTS) <= declared


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(_lib.MdhFrameInfo) == 56 and _lib.FRAME_INFO_DTYPE.itemsize == 56
    assert C.sizeof(_lib.MdhCfg) == 16 * 4 + 8 + 8 and _lib.MdhCfg.struct_size.offset == 0 and _lib.MdhCfg.mask_area.offset == 64
    assert C.sizeof(_lib.MdhCollectorCfg) == 8 + 11 * 8 + 4 * 4 and _lib.MdhCollectorCfg.struct_size.offset == 0
    assert re.search(r"#define MDH_ABI_VERSION\s+%d\b" % _lib.MDH_ABI_VERSION, open(os.path.join(ROOT, "include", "metdet_hip.h")).read())
    assert _lib.load().mdh_abi_version() == _lib.MDH_ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "metdet_hip.h")).read()
    assert "#define MDH_MAX_LINES 512" in hdr and _lib.MDH_MAX_LINES == 512


def test_wrong_struct_size_is_refused():
    """ABI guard: a caller built against another header (round 1/2's mdh_cfg had no struct_size: its first field was `detector`,
    0 or 1) is refused before any other field is read -- no GPU needed, the check comes first."""
    import ctypes as C
    L = _lib.load()
    cfg = _lib.MdhCfg()
    cfg.sensitivity, cfg.interval, cfg.mask_area, cfg.hough_threshold = 1, 2, 100.0, 10
    cfg.roi[2], cfg.roi[3] = 8, 8
    for bad in (0, 1, 72, 76, 84):
        cfg.struct_size = bad
        assert not L.mdh_create(0, 16, 16, 4, C.byref(cfg))
        assert "struct_size" in _lib.last_error() and "80" in _lib.last_error()
    ccfg = _lib.MdhCollectorCfg()
    ccfg.fps, ccfg.num_cls, ccfg.exp_frame, ccfg.runtime_w, ccfg.runtime_h = 25.0, 10, 1, 960, 540
    ccfg.struct_size = 104
    assert not L.mdh_collector_create(C.byref(ccfg))
    assert "struct_size" in _lib.last_error()
    ccfg.struct_size = C.sizeof(_lib.MdhCollectorCfg)
    h = L.mdh_collector_create(C.byref(ccfg))
    assert h
    L.mdh_collector_destroy(h)


def test_integration_md_stub_matches_the_library():
    """INTEGRATION.md option B is what a maintainer pastes into the reference: its ctypes structs are executed here and compared,
    field by field, with the binding the tests exercise (round 2's stub ended at mask_area: 72 bytes against the library's 80)."""
    import ctypes as C
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class _Cfg(C.Structure)"):doc.index("_L.mdh_create.restype")]
    block = block[:block.index("assert _L.")] if "assert _L." in block else block
    ns = {"C": C}
    exec(block, ns)
    for doc_cls, lib_cls in ((ns["_Cfg"], _lib.MdhCfg), (ns["_Info"], _lib.MdhFrameInfo)):
        assert C.sizeof(doc_cls) == C.sizeof(lib_cls)
        assert [(n, getattr(doc_cls, n).offset, getattr(doc_cls, n).size) for n, _ in doc_cls._fields_] == \
               [(n, getattr(lib_cls, n).offset, getattr(lib_cls, n).size) for n, _ in lib_cls._fields_]
    assert "struct_size=C.sizeof(_Cfg)" in doc and f"mdh_abi_version() == {_lib.MDH_ABI_VERSION}" in doc


def test_inject_sums_contract_is_documented():
    hdr = open(os.path.join(ROOT, "include", "metdet_hip.h")).read()
    assert "mdh_hw_queues_hint" in hdr and _lib.load().mdh_hw_queues_hint() >= 0


def test_frame_info_field_order_of_the_per_frame_path():
    """M3Detector.detect() unpacks Context.detect_fields()'s tuple positionally (one tolist() per frame instead of a record copy
    and a lookup per attribute): the order it assumes is FRAME_INFO_DTYPE's, which is mdh_frame_info's (include/metdet_hip.h)."""
    assert _lib.FRAME_INFO_DTYPE.names == ("n_raw", "n_nms", "lit_count", "bi_threshold", "bi_threshold_float", "snr", "sigma",
                                           "dst_sum", "hough_gap")
    assert [f[0] for f in _lib.MdhFrameInfo._fields_] == list(_lib.FRAME_INFO_DTYPE.names)
    rec = np.zeros(1, _lib.FRAME_INFO_DTYPE)
    rec[0] = (3, 2, 40, 11, 10.7, 2.5, 1.25, 0.01, 8.0)
    assert rec.tolist()[0] == (3, 2, 40, 11, 10.7, 2.5, 1.25, 0.01, 8.0)


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: without a visible GPU the constructor raises instead of silently computing elsewhere."""
    import ctypes as C
    L = _lib.load()
    cfg = _lib.MdhCfg()
    cfg.sensitivity, cfg.interval, cfg.mask_area = 1, 2, 100.0
    cfg.roi[2], cfg.roi[3] = 8, 8
    cfg.hough_threshold = 10
    h = L.mdh_create(0, 16, 16, 4, C.byref(cfg))
    if h:  # a GPU is present (this test also runs on the GPU box): just clean up
        L.mdh_destroy(h)
    else:
        assert "GPU" in _lib.last_error() or "HIP" in _lib.last_error() or "hip" in _lib.last_error()


def test_product_does_not_import_oracle():
    for name in os.listdir(os.path.join(ROOT, "metdetpy_amd")):
        if name.endswith(".py"):
            src = open(os.path.join(ROOT, "metdetpy_amd", name)).read()
            assert "oracle" not in src.replace("# oracle", ""), f"{name} mentions the oracle"


def test_product_and_bench_import_no_torch():
    """north_star: "no PyTorch".  The package, bench.py and bench_modes.py do not import torch; the launcher the driver uses
    (`python -m torch.distributed.run`) only exports RANK / WORLD_SIZE / MASTER_*, which rendezvous.py reads."""
    files = [os.path.join(ROOT, "metdetpy_amd", f) for f in os.listdir(os.path.join(ROOT, "metdetpy_amd")) if f.endswith(".py")]
    files += [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "bench_modes.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for path in files:
        src = open(path).read()
        assert not re.search(r"^\s*(import torch|from torch)", src, re.M), f"{path} imports torch"
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, sys.argv[1]); import metdetpy_amd, metdetpy_amd.dist, "
                            "metdetpy_amd.bands, metdetpy_amd.timeshard, metdetpy_amd.rccl, metdetpy_amd.rendezvous, bench, bench_modes; "
                            "assert 'torch' not in sys.modules", ROOT], capture_output=True, text=True)
    assert probe.returncode == 0, probe.stderr[-1500:]


def test_g3_std_roi_host(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "g3_std_roi.json"))):
        h, w, kind = c["h"], c["w"], c["kind"]
        m = np.ones((h, w), np.uint8)
        if kind == "bottom":
            m[int(h * 0.62):] = 0
        elif kind == "bottom_heavy":
            m[int(h * 0.35):] = 0
        elif kind == "corner":
            m[int(h * 0.5):, :int(w * 0.5)] = 0
        elif kind == "stripe":
            m[:, int(w * 0.45):int(w * 0.5)] = 0
        assert list(U.select_std_roi(m, c["area"])) == c["roi"], c
    with pytest.raises(ValueError):
        U.select_std_roi(np.ones((10, 10), np.uint8), 0)


def test_g4_native_lineset_nms(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "g4_lineset_nms.json"))):
        out, prob = U.lineset_nms(np.array(c["lines"], np.int32))
        assert out.tolist() == c["out"]
        assert np.array_equal(prob, np.array(c["nonline"], np.float64))
    out, prob = U.lineset_nms(np.zeros((0, 4), np.int32))
    assert len(out) == 0 and len(prob) == 0
    # documented tie rule: equal lengths -> the later index leads
    out, _ = U.lineset_nms(np.array([[0, 0, 10, 0], [100, 100, 110, 100]], np.int32))
    assert out.tolist() == [[100, 100, 110, 100], [0, 0, 10, 0]]


def test_config_parsing_like_reference_json():
    cfg = {
        "name": "M3Detector", "window_sec": 1,
        "cfg": {"binary": {"adaptive_bi_thre": True, "init_value": 7, "sensitivity": "normal", "area": 0.1, "interval": 2},
                "hough_line": {"threshold": 10, "min_len": 10, "max_gap": 10},
                "dynamic": {"dy_mask": True, "window_sec": 5}}}
    d = DetectorCfg.from_dict(cfg)
    assert isinstance(d.cfg, BinaryCfg) and d.cfg.hough_line.max_gap == 10 and d.cfg.binary.sensitivity == "normal"
    assert d.to_dict() == cfg
    col = {"meteor_cfg": {"min_len": 15, "max_interval": 4, "time_range": [0, 8], "speed_range": [2, 21],
                          "drct_range": [0, 0.6], "det_thre": 0.5, "thre2": 2048},
           "recheck_cfg": {"switch": False, "model": {"name": "YOLOModel"}, "save_path": ""},
           "positive_cfg": {"positive_cates": ["METEOR", "RED_SPRITE"]}}
    c = CollectorCfg.from_dict(col)
    assert c.meteor_cfg.thre2 == 2048 and c.recheck_cfg.switch is False


def test_stream_sharding():
    from metdetpy_amd.dist import shard_streams
    for world in (1, 2, 3, 8):
        for n in (1, 5, 8, 17):
            seen = sorted(s for r in range(world) for s in shard_streams(n, r, world))
            assert seen == list(range(n))


GATHER_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
from metdetpy_amd import dist as md
from tests.gloo_group import make_group
group = make_group(sys.argv[2])
rank, world = group.rank, group.world
streams = md.shard_streams(int(sys.argv[3]), rank, world)
local_recs = {s: [{"stream": s, "rank": rank, "pt1": [s, rank], "score": 0.5 + s}] * (s % 3) for s in streams}
merged = md.gather_records(local_recs, group)
tmax = md.max_over_ranks(1.0 + rank, group)
counts = group.all_gather_i64([rank, 10 * rank, -rank])
assert counts == [[r, 10 * r, -r] for r in range(world)], counts
if rank == 0:
    print(json.dumps({"merged": {str(k): v for k, v in merged.items()}, "tmax": tmax, "backend": group.backend}))
else:
    assert merged is None
group.close()
"""


@pytest.mark.parametrize("kind,world,streams", [("store", 2, 5), ("gloo", 2, 5), ("store", 8, 8), ("store", 8, 19)])
def test_multi_process_final_gather(tmp_path, kind, world, streams):
    """Independent streams, one (or more) per rank: the final gather of the detection records to rank 0, the max-over-ranks time and
    the all-gather of the counts -- over the product's store-backed group (no GPU here: the RCCL branch of the same calls runs in
    tests/test_gpu_bench_contract.py) and over gloo."""
    from tests.mp import run_ranks
    res = run_ranks(tmp_path, "gather_worker", GATHER_WORKER, world, args=(kind, streams), timeout=300)
    assert res["tmax"] == float(world) and res["backend"] == kind
    assert sorted(int(k) for k in res["merged"]) == list(range(streams))
    for k, v in res["merged"].items():
        assert len(v) == int(k) % 3 and all(r["stream"] == int(k) and r["rank"] == int(k) % world for r in v)


def test_rendezvous_store_survives_a_stale_port_file(tmp_path, monkeypatch):
    """A crashed run leaves its port file behind: a new job on the same MASTER_PORT must not trust it (handshake nonce)."""
    import struct
    import threading
    from metdetpy_amd import rendezvous as rv
    from tests.mp import free_port
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None  # (re-read TMPDIR)
    try:
        port = free_port()
        dead = free_port()
        with open(rv._port_file("127.0.0.1", port), "wb") as f:
            f.write(struct.pack("<I", dead) + b"x" * 16)  # nobody listens there
        box = {}

        def rank1():
            g = rv.Group(1, 2, "127.0.0.1", port, timeout_s=20)
            box["got"] = g.broadcast_bytes(None, 0)
            box["ag"] = g.all_gather_bytes(b"one")
            g.close()

        th = threading.Thread(target=rank1)
        th.start()
        import time
        time.sleep(0.3)  # rank 1 is already retrying against the stale file
        g0 = rv.Group(0, 2, "127.0.0.1", port, timeout_s=20)
        g0.broadcast_bytes(b"uid", 0)
        assert g0.all_gather_bytes(b"zero") == [b"zero", b"one"]
        g0.close()
        th.join(30)
        assert box == {"got": b"uid", "ag": [b"zero", b"one"]}
        assert not g0.store.server.data, "the store kept keys of finished collectives"
        with pytest.raises(rv.RendezvousError):
            rv.Group(1, 2, "127.0.0.1", free_port(), timeout_s=0.5)  # no rank 0 anywhere
    finally:
        tempfile.tempdir = None


def _two_threads(port, body0, body1, timeout_s=20):
    """Ranks 0 and 1 of a store-backed group as two threads of this process; returns {rank: result or exception}."""
    import threading
    from metdetpy_amd import rendezvous as rv
    box = {}

    def run(rank, body):
        try:
            g = rv.Group(rank, 2, "127.0.0.1", port, timeout_s=timeout_s)
            try:
                box[rank] = body(g)
            finally:
                g.close()
        except BaseException as e:  # noqa: BLE001
            box[rank] = e

    th = [threading.Thread(target=run, args=(r, b)) for r, b in ((0, body0), (1, body1))]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    return box


def test_rccl_unique_id_failure_on_rank_0_reaches_every_rank(monkeypatch):
    """rank 0 failing to make the ncclUniqueId (no usable librccl) must not leave it one collective behind the
    others -- every rank gets the reason at once, and the agreement on the store fall-back that follows still works."""
    from metdetpy_amd import rccl
    from tests.mp import free_port

    def boom():
        raise OSError("librccl.so: cannot open shared object file")
    monkeypatch.setattr(rccl, "new_unique_id", boom)

    def body(g):
        try:
            rccl.Communicator.from_group(g, device=0, timeout_s=5)
            err = None
        except rccl.RcclError as e:
            err = str(e)
        return err, g.all_gather_object(err)  # what dist.Group does next: the outcomes, through the same control plane

    import time
    t0 = time.monotonic()
    box = _two_threads(free_port(), body, body, timeout_s=10)
    assert time.monotonic() - t0 < 8, "the ranks waited for a time-out instead of getting the answer"
    for r in (0, 1):
        assert not isinstance(box[r], BaseException), box[r]
        err, outcomes = box[r]
        assert err and "cannot open shared object file" in err
        assert outcomes == [box[0][0], box[1][0]]


def test_rendezvous_unpickles_only_what_carries_the_jobs_seal():
    """The object collectives pickle; a value that was not sealed by a rank of this job (somebody else's SET on the store's port) is
    refused before pickle.loads sees it.  A stranger without the secret does not even get through the handshake."""
    import pickle
    import socket
    from metdetpy_amd import rendezvous as rv
    from tests.mp import free_port
    port = free_port()
    hit = []

    class Evil:
        def __reduce__(self):
            return (hit.append, ("unpickled",))

    def body0(g):
        # a forged value for the NEXT collective's key of rank 1, written through rank 0's own (authenticated) connection: the worst case
        g.store.set(f"c{g.seq + 1}/1/v", pickle.dumps(Evil(), protocol=4), reads=1)
        try:
            g.all_gather_object("zero")
        except rv.RendezvousError as e:
            return str(e)
        return "accepted"

    def body1(g):
        import time
        time.sleep(1.0)  # (rank 0 reads the forged value first; this rank's own SET then overwrites a key nobody waits for)
        try:
            return g.all_gather_object("one")
        except rv.RendezvousError as e:
            return str(e)

    box = _two_threads(port, body0, body1, timeout_s=5)
    assert "does not carry this job's seal" in str(box[0]) and not hit
    # the handshake: a client that does not know the secret is turned away, and never learns it from the exchange
    srv = rv._Server("127.0.0.1", 0, b"s" * 16)
    try:
        with pytest.raises(rv.RendezvousError):
            rv.Store._connect("127.0.0.1", srv.port, b"x" * 16)
        s = socket.create_connection(("127.0.0.1", srv.port), timeout=5)
        s.sendall(rv._MAGIC + b"c" * 16)
        reply = rv._recv_exact(s, 48)
        assert b"s" * 16 not in reply
        s.sendall(b"\0" * 32)
        assert rv._recv_exact(s, 1) == b"\0"
        s.close()
    finally:
        srv.close()


def test_rendezvous_fixed_port_needs_a_secret_and_port_files_are_private(tmp_path, monkeypatch):
    import stat
    import tempfile
    from metdetpy_amd import rendezvous as rv
    from tests.mp import free_port
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    tempfile.tempdir = None
    try:
        monkeypatch.setenv("MDH_RDV_PORT", str(free_port()))
        monkeypatch.delenv("MDH_RDV_SECRET", raising=False)
        with pytest.raises(rv.RendezvousError, match="MDH_RDV_SECRET"):
            rv.Group(0, 1, "127.0.0.1", free_port(), timeout_s=2)
        with pytest.raises(rv.RendezvousError, match="MDH_RDV_SECRET"):
            rv.Group(1, 2, "127.0.0.1", free_port(), timeout_s=2)
        monkeypatch.setenv("MDH_RDV_SECRET", "tonight's job")
        g = rv.Group(0, 1, "127.0.0.1", free_port(), timeout_s=2)
        assert g.all_gather_object({"a": 1}) == [{"a": 1}]
        g.close()
        monkeypatch.delenv("MDH_RDV_PORT")
        port = free_port()
        g = rv.Group(0, 1, "127.0.0.1", port, timeout_s=2)
        f = rv._port_file("127.0.0.1", port)
        assert stat.S_IMODE(os.stat(f).st_mode) == 0o600 and stat.S_IMODE(os.stat(os.path.dirname(f)).st_mode) == 0o700
        g.close()
        os.chmod(os.path.dirname(f), 0o755)  # a directory others could have prepared is refused
        with pytest.raises(rv.RendezvousError, match="private directory"):
            rv.Group(0, 1, "127.0.0.1", port, timeout_s=2)
    finally:
        tempfile.tempdir = None


def test_rendezvous_connection_is_usable_after_a_timeout():
    """a GET that timed out used to leave its request parked on the connection -- every later call on the same
    Store blocked and failed.  The client reconnects: the recovery paths (agreeing on the store fall-back, the barrier of close())
    keep working."""
    from metdetpy_amd import rendezvous as rv
    from tests.mp import free_port
    g = rv.Group(0, 2, "127.0.0.1", free_port(), timeout_s=0.5)
    with pytest.raises(rv.RendezvousError):
        g.store.get("never/set")
    g.store.set("k", b"v")
    assert g.store.get("k") == b"v"
    assert g.store.add("n", 5) == 5
    g.store.close()


NO_FALLBACK_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import bench
from metdetpy_amd import _lib, rccl

class FakeLib:  # a node with a GPU per rank ...
    def mdh_device_count(self): return 2
_lib.load = lambda: FakeLib()
def no_rccl(*a, **k): raise rccl.RcclError("ncclCommInitRank: unhandled system error")   # ... on which RCCL does not come up
rccl.Communicator.from_group = classmethod(lambda cls, *a, **k: no_rccl())
try:
    D = bench.Dist()
    out = {"exit": None, "backend": D.backend}
    D.close()
except SystemExit as e:
    out = {"exit": e.code}
print(json.dumps(out))
"""


@pytest.mark.parametrize("forced_store", [False, True])
def test_bench_refuses_the_store_fallback_when_every_rank_has_a_gpu(tmp_path, forced_store):
    """N distinct devices visible and RCCL cannot form -> `bench.py --gpus N` fails (exit code 3) on every
    rank instead of quietly running its collectives through the rendezvous store; MDH_DIST_BACKEND=store still asks for the store."""
    from tests.mp import run_ranks
    res = run_ranks(tmp_path, "no_fallback", NO_FALLBACK_WORKER, 2, env={"MDH_DIST_BACKEND": "store"} if forced_store else {"MDH_RCCL_TIMEOUT": "20"}, timeout=120)
    assert res == ({"exit": None, "backend": "store"} if forced_store else {"exit": 3})


def test_profiles_readme_table_matches_the_tracked_csv():
    """profiles/README.md quoted an AverageNs the tracked CSV did not hold.  The README's table of the
    round's `*_depth1.csv` summaries is generated (tools/profiles_table.py) and checked here against the files themselves: the
    duration as the CSV has it, the bytes of bench.compulsory_bytes, the fraction within 1 %; DESIGN.md may quote a file's
    fraction only as the README has it."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import profiles_table as pt
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "profiles", "README.md")).read()
    listed = pt.parse(text)
    assert listed, "profiles/README.md holds no generated table"
    tag = listed[0][0].split("_")[0]
    now = {r[0]: r for r in pt.rows(tag)}
    assert {f for f, *_ in listed} == set(now), (sorted(f for f, *_ in listed), sorted(now))
    for f, avg, nbytes, frac in listed:
        r = now[f]
        assert abs(avg - r[3]) <= 0.5 and nbytes == r[4], (f, avg, r[3], nbytes, r[4])
        assert abs(frac - r[6]) <= 0.01 * r[6] + 0.0006, (f, frac, r[6])
    design = open(os.path.join(root, "DESIGN.md")).read()
    for f, avg, nbytes, frac in listed:  # every "<file> ... 0.xxx" quote of DESIGN.md in the same sentence must be the README's figure
        for m in re.finditer(re.escape(f) + r"[^.\n]{0,120}?\b(0\.\d{3})\b", design):
            assert abs(float(m.group(1)) - frac) <= 0.01 * frac + 0.0006, (f, m.group(1), frac)


def test_every_environment_switch_is_documented():
    """The library and the bench read MDH_* switches from the environment (A/B builds of one path, experiments, instrumentation);
    INTEGRATION.md's table is where a maintainer finds out what each does: none may be missing from it."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for path in glob.glob(os.path.join(root, "metdetpy_amd", "csrc", "*.h*")) + glob.glob(os.path.join(root, "metdetpy_amd", "*.py")) + [
            os.path.join(root, "bench.py"), os.path.join(root, "bench_modes.py")]:
        text = open(path).read()
        names |= set(re.findall(r'getenv\("(MDH_[A-Z0-9_]+)"', text))
        names |= set(re.findall(r'environ(?:\.get\(|\[)"(MDH_[A-Z0-9_]+)"', text))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
