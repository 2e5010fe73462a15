"""CPU suite, part 3: the C-ABI library builds for gfx950, loads, exports every symbol the
header declares, and rejects bad arguments before touching the GPU.  No compute calls."""
import ctypes
import os
import re
import subprocess

import pytest

import reflectance_filtering_amd as rf
from reflectance_filtering_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(name="reflectance_filtering.h"):
    with open(os.path.join(ROOT, "include", name)) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    lib = _ffi.load_library()
    names = _header_functions()
    assert set(names) == set(_ffi.EXPORTS)
    for name in names:
        assert getattr(lib, name) is not None
    assert lib.rf_version() == 103
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH]).decode()
    for name in names:
        assert re.search(r"\bT %s\b" % name, out), name
    # the test / benchmark switches are declared in their own header, outside the boundary
    dbg = _header_functions("reflectance_filtering_debug.h")
    assert set(dbg) == set(_ffi.DEBUG_EXPORTS)
    for name in dbg:
        assert re.search(r"\bT %s\b" % name, out), name
    exported = set(re.findall(r"\bT (rf_[a-z0-9_]+)\b", out))
    assert exported == set(names) | set(dbg), exported ^ (set(names) | set(dbg))


def test_build_records_its_toolchain(built, record_property, capsys):
    """The machine-code audits below (registers touched while a read is in flight, packed-FMA
    operand routing) hold for the code ONE compiler wrote: the library says which, the log shows
    it, and a library built by something else than the hipcc on this machine is named as such."""
    lib = _ffi.load_library()
    info = lib.rf_debug_build_info().decode()
    record_property("toolchain", info)
    with capsys.disabled():
        print("\nlibrf_hip.so built with: %s" % info)
    assert "HIP version" in info and "clang" in info, info
    hipcc = "/opt/rocm/bin/hipcc"
    if os.path.exists(hipcc):
        here = subprocess.check_output([hipcc, "--version"]).decode().splitlines()[:2]
        assert all(line.replace('"', "").replace("'", "") in info for line in here), (
            "the library was built with another toolchain than this machine's: %s" % info)


def test_code_object_is_gfx950_only(built, tmp_path):
    tool = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(tool):
        pytest.skip("llvm-objdump not available")
    import shutil
    copy = str(tmp_path / "librf_hip.so")   # --offloading extracts the bundles next to its input
    shutil.copy(_ffi.LIB_PATH, copy)
    out = subprocess.check_output([tool, "--offloading", copy],
                                  stderr=subprocess.STDOUT).decode()
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_no_packed_f32_op_sel_on_scalar_operands(built, tmp_path):
    """gfx950 hazard found while tuning the CNN: v_pk_fma_f32 with an SGPR-pair
    source whose halves are re-routed by op_sel / op_sel_hi gives wrong results; hipcc emits
    exactly that for a splat of an SGPR element.  No kernel in the library may contain it."""
    tool = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(tool):
        pytest.skip("llvm-objdump not available")
    import glob
    import shutil
    copy = str(tmp_path / "librf_hip.so")
    shutil.copy(_ffi.LIB_PATH, copy)
    subprocess.check_output([tool, "--offloading", copy], stderr=subprocess.STDOUT)
    bad, seen = [], 0
    for obj in glob.glob(copy + ".*gfx950"):
        text = subprocess.check_output([tool, "-d", obj]).decode()
        for line in text.splitlines():
            m = re.search(r"\b(v_pk_(?:fma|mul|add)_f32)\s+([^/]*)", line)
            if not m:
                continue
            seen += 1
            ops = m.group(2)
            srcs = [o.strip() for o in ops.split(" op_sel")[0].split(",")][1:]
            sel = re.search(r"op_sel:\[([0-9,]+)\]", ops)
            sel_hi = re.search(r"op_sel_hi:\[([0-9,]+)\]", ops)
            sel = [int(v) for v in sel.group(1).split(",")] if sel else [0] * len(srcs)
            sel_hi = [int(v) for v in sel_hi.group(1).split(",")] if sel_hi else [1] * len(srcs)
            for i, src in enumerate(srcs):
                if src.startswith("s[") and i < len(sel) and (sel[i] != 0 or sel_hi[i] != 1):
                    bad.append(line.strip())
    assert seen > 0, "expected packed FMAs in the CNN kernel"
    assert not bad, bad[:5]


def _disassembly(tmp_path_factory):
    tool = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(tool):
        pytest.skip("llvm-objdump not available")
    import glob
    import shutil
    d = tmp_path_factory.mktemp("disasm")
    copy = str(d / "librf_hip.so")
    shutil.copy(_ffi.LIB_PATH, copy)
    subprocess.check_output([tool, "--offloading", copy], stderr=subprocess.STDOUT)
    funcs = {}
    for obj in glob.glob(copy + ".*gfx950"):
        name = None
        for line in subprocess.check_output([tool, "-d", obj]).decode().splitlines():
            m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
            if m:
                name = m.group(2)
                funcs[name] = []
                _FLOW[name] = {"start": int(m.group(1), 16), "addr": [], "target": []}
                continue
            m = re.match(r"^\s+([a-z_0-9]+)\s*([^/]*)(?://\s*([0-9A-Fa-f]+):)?", line)
            if m and name is not None:
                funcs[name].append((m.group(1), m.group(2).strip()))
                t = re.search(r"<[^>]*\+0x([0-9a-f]+)>\s*$", line)
                _FLOW[name]["addr"].append(int(m.group(3), 16) if m.group(3) else None)
                _FLOW[name]["target"].append(int(t.group(1), 16) if t else None)
    return funcs


_FLOW = {}   # per function: start address, address of every instruction, branch targets (offsets)


@pytest.fixture(scope="module")
def disassembly(built, tmp_path_factory):
    return _disassembly(tmp_path_factory)


def _sgprs(text):
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return regs


def _smem_hazards(name, insts, flow):
    """Walk the control-flow graph of one function with the set of SGPRs that scalar loads in flight will
    write (scalar loads return out of order: only `lgkmcnt(0)` completes them) and fail on any instruction
    that touches one of them.  Returns the number of scalar loads seen."""
    checked = 0
    index_of = {a: i for i, a in enumerate(flow["addr"]) if a is not None}
    seen = set()
    work = [(0, frozenset())]
    while work:
        i, pending = work.pop()
        while i < len(insts):
            if (i, pending) in seen:
                break
            seen.add((i, pending))
            op, args = insts[i]
            if op.startswith("s_load_dword"):
                dst, _, rest = args.partition(",")
                assert not (_sgprs(rest) & pending), (name, op, args)
                pending = pending | frozenset(_sgprs(dst))
                checked += 1
            elif op == "s_waitcnt":
                if "lgkmcnt(0)" in args:
                    pending = frozenset()
            elif op in ("s_endpgm", "s_setpc_b64"):
                break
            elif op.startswith("s_cbranch") or op == "s_branch":
                target = flow["target"][i]
                assert target is not None and flow["start"] + target in index_of, (name, op, args)
                work.append((index_of[flow["start"] + target], pending))
                if op == "s_branch":
                    break
            else:
                assert not (_sgprs(args) & pending), (name, op, args, sorted(pending)[:4])
            i += 1
    return checked


def test_inline_asm_scalar_loads_are_not_touched_before_their_wait(disassembly):
    """The CNN kernels issue s_load_dwordx16 / x2 in one inline-asm statement and wait for them in a
    later one, the bilateral's tap loops their weight windows (s_load_dwordx8) and the next row's
    half-width; the compiler does not know the loads are in flight, so nothing it places in between may
    read or write their destination SGPRs (it could copy or spill stale values).  Checked on the
    machine code of every kernel that streams this way, along every path of the control-flow graph."""
    checked = 0
    for name, insts in disassembly.items():
        if "cnn_reflectance" not in name and "jbf_tile64" not in name and "jbf_slab" not in name:
            continue
        checked += _smem_hazards(name, insts, _FLOW[name])
    assert checked > 20, "expected the weight-streaming scalar loads"


def _vgprs(text):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return regs


def _lds_hazards(name, insts, flow):
    """Walk the control-flow graph of one function with the queue of LDS reads in flight (destination
    register sets, oldest first) and fail on any instruction that touches a register whose read has not
    been waited for.  Returns the number of LDS reads seen."""
    checked = 0
    index_of = {a: i for i, a in enumerate(flow["addr"]) if a is not None}
    seen = set()
    work = [(0, ())]
    while work:
        i, pending = work.pop()
        while i < len(insts):
            key = (i, pending)
            if key in seen:
                break
            seen.add(key)
            op, args = insts[i]
            busy = frozenset().union(*pending) if pending else frozenset()
            if op.startswith("ds_"):     # every LDS operation takes a place in the in-order queue
                reads = op.startswith("ds_read") or op.startswith("ds_load") or "_rtn" in op
                dst = frozenset(_vgprs(args.split(",")[0])) if reads else frozenset()
                # (its destination may be that of an older read: returns are in order)
                srcs = args.split(",", 1)[1] if reads and "," in args else args
                assert not (_vgprs(srcs) & busy), (name, op, args)
                pending = pending + (dst,)
                checked += 1 if reads else 0
            elif op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", args)
                if m:
                    keep = int(m.group(1))
                    pending = pending[len(pending) - keep:] if 0 < keep < len(pending) else \
                        (() if keep == 0 else pending)
            elif op in ("s_endpgm", "s_setpc_b64"):
                assert not pending, (name, op, "LDS reads still in flight at the end")
                break
            elif op.startswith("s_cbranch") or op == "s_branch":
                target = flow["target"][i]
                assert target is not None and flow["start"] + target in index_of, (name, op, args)
                work.append((index_of[flow["start"] + target], pending))
                if op == "s_branch":
                    break
            elif op != "s_barrier":
                assert not (_vgprs(args) & busy), (name, op, args, sorted(busy)[:8])
            # (the counter has 4 bits: with 15 operations outstanding the next one waits for the oldest;
            #  a write at the head of the queue pins no register)
            pending = pending[-15:]
            while pending and not pending[0]:
                pending = pending[1:]
            i += 1
    return checked


def test_lds_reads_in_flight_are_not_touched_before_their_wait(disassembly):
    """The joint bilateral's asm tap loops issue LDS reads in one inline-asm statement and wait for
    them in a later one - the round-5 loop keeps four gathers in flight ACROSS a column step
    (`s_waitcnt lgkmcnt(4)`), across the groups of a row and across rows - and the compiler does not
    know they are in flight: nothing between a `ds_read*` and the wait that covers it may read or write
    its destination VGPRs (a register move would copy a stale value, a re-use would be overwritten
    when the data lands).  LDS operations of a wave return in order, so `lgkmcnt(n)` leaves at most the
    n youngest reads pending (scalar loads share the counter and only make a wait stricter).  Checked
    on the machine code of every joint-bilateral kernel along every path of its control-flow graph
    (both sides of every branch, loop back edges included: the compiler writes a row loop's header
    itself), function ends included (no read may still be pending at `s_endpgm`)."""
    checked = kernels = 0
    for name, insts in disassembly.items():
        if "jbf_" not in name or "f32" in name:
            continue
        kernels += 1
        checked += _lds_hazards(name, insts, _FLOW[name])
    assert kernels >= 8 and checked > 2000, (kernels, checked)


def test_the_lds_hazard_check_follows_back_edges():
    """The checker on hand-made code: a loop that leaves a gather in flight across its back edge is fine
    as long as the loop's first instructions keep off the gather's register, and is caught when one of
    them copies it - which a reading of the text from top to bottom would miss (nothing is in flight the
    first time round)."""
    def function(first):
        insts = [first,
                 ("ds_read_b32", "v1, v2"),
                 ("s_waitcnt", "lgkmcnt(1)"),
                 ("s_cmp_lg_u32", "s0, 0"),
                 ("s_cbranch_scc1", "65531"),
                 ("s_waitcnt", "lgkmcnt(0)"),
                 ("v_mul_f32_e32", "v3, v1, v1"),
                 ("s_endpgm", "")]
        flow = {"start": 0x100, "addr": [0x100 + 4 * i for i in range(len(insts))],
                "target": [None, None, None, None, 0, None, None, None]}
        return insts, flow
    assert _lds_hazards("ok", *function(("v_add_u32_e32", "v5, v5, v6"))) >= 1
    with pytest.raises(AssertionError):
        _lds_hazards("bad", *function(("v_mov_b32_e32", "v7, v1")))
    # the scalar-load walk on the same skeleton: a window requested at the loop's end, copied at its top
    def scalar(first):
        insts = [first,
                 ("s_load_dwordx2", "s[4:5], s[0:1], 0x0"),
                 ("s_cmp_lg_u32", "s8, 0"),
                 ("s_cbranch_scc1", "65532"),
                 ("s_waitcnt", "lgkmcnt(0)"),
                 ("s_endpgm", "")]
        flow = {"start": 0x100, "addr": [0x100 + 4 * i for i in range(len(insts))],
                "target": [None, None, None, 0, None, None]}
        return insts, flow
    assert _smem_hazards("ok", *scalar(("s_add_i32", "s8, s8, -1"))) >= 1
    with pytest.raises(AssertionError):
        _smem_hazards("bad", *scalar(("s_mov_b32", "s9, s4")))
    with pytest.raises(AssertionError):      # ... and a read still in flight at the end
        insts, flow = function(("v_add_u32_e32", "v5, v5, v6"))
        insts[5] = ("s_nop", "0")
        _lds_hazards("end", insts, flow)


def test_dpp_reads_respect_the_valu_write_hazard(disassembly):
    """gfx9 needs two wait states between a VALU write of a VGPR and a DPP read of it; the hazard
    recogniser does not look inside inline asm (the guided filter's last scan step is a hand-placed
    s_nop + v_add_u32_dpp run).  Checked for every DPP instruction of the library."""
    seen = 0
    for name, insts in disassembly.items():
        recent = [set(), set()]                      # VGPRs written by the last two issue slots
        for op, args in insts:
            if op == "s_nop":
                for _ in range(int(args.split()[0], 0) + 1):
                    recent = [recent[1], set()]
                continue
            ops = [a.strip() for a in args.split(",")]
            if "_dpp" in op or " row_" in args or "quad_perm" in args or "wave_" in args:
                seen += 1
                src0 = ops[1].split()[0] if len(ops) > 1 else ""
                regs = set()
                m = re.match(r"v\[(\d+):(\d+)\]", src0)
                if m:
                    regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
                elif re.match(r"v(\d+)$", src0):
                    regs = {int(src0[1:])}
                assert not (regs & (recent[0] | recent[1])), (name, op, args)
            written = set()
            if op.startswith("v_") and not op.startswith("v_cmp") and ops and ops[0]:
                m = re.match(r"v\[(\d+):(\d+)\]", ops[0])
                if m:
                    written = set(range(int(m.group(1)), int(m.group(2)) + 1))
                elif re.match(r"v(\d+)$", ops[0]):
                    written = {int(ops[0][1:])}
            recent = [recent[1], written]
    assert seen > 50, "expected the DPP scans of the guided filter's stage 1"


def test_argument_validation_needs_no_gpu(built):
    lib = _ffi.load_library()
    bufs = [ctypes.create_string_buffer(64 * 64 * 3) for _ in range(3)]
    p, q, o = (ctypes.cast(b, ctypes.c_void_p) for b in bufs)
    assert lib.rf_jbf_u8(None, q, o, 1, 4, 4, 3, 3, -1, 20.0, 22.0, 4, 0, None) == _ffi.RF_E_BADARG
    assert b"NULL" in lib.rf_last_error()
    assert lib.rf_jbf_u8(p, q, o, 1, 4, 4, 2, 3, -1, 20.0, 22.0, 4, 0, None) == _ffi.RF_E_UNSUPPORTED
    assert lib.rf_jbf_u8(p, q, o, 1, 0, 4, 3, 3, -1, 20.0, 22.0, 4, 0, None) == _ffi.RF_E_BADARG
    assert lib.rf_jbf_u8(p, q, o, 1, 4, 4, 3, 3, -1, 20.0, 22.0, 9, 0, None) == _ffi.RF_E_UNSUPPORTED
    # undocumented flag bits are rejected (the old benchmark bits moved behind rf_debug_option)
    assert lib.rf_jbf_u8(p, q, o, 1, 4, 4, 3, 3, -1, 20.0, 22.0, 4, 0x1000, None) == _ffi.RF_E_BADARG
    assert b"flag" in lib.rf_last_error()
    # overlap is checked by range, not by pointer equality
    big = ctypes.create_string_buffer(256)
    base = ctypes.cast(big, ctypes.c_void_p).value
    assert lib.rf_jbf_u8(base, base + 128, base + 140, 1, 4, 4, 3, 3, -1, 20.0, 22.0, 4, 0,
                         None) == _ffi.RF_E_BADARG
    assert b"overlap" in lib.rf_last_error()
    assert lib.rf_gf_u8(base, base + 64, base + 80, 1, 4, 4, 3, 3, 2, 1.0, 1, p, 1 << 20,
                        None) == _ffi.RF_E_BADARG
    assert lib.rf_debug_option(b"no_such_option", 1) == _ffi.RF_E_BADARG
    assert lib.rf_debug_option(b"gf_two_kernel", 1) == 0
    assert lib.rf_debug_option(b"gf_two_kernel", 0) == 1
    assert lib.rf_gf_u8(p, q, o, 1, 4, 4, 1, 3, 2, 1.0, 1, p, 1 << 20, None) == _ffi.RF_E_UNSUPPORTED
    assert lib.rf_gf_u8(p, q, o, 1, 4, 4, 3, 3, 2, 1.0, 0, p, 1 << 20, None) == _ffi.RF_E_BADARG
    assert lib.rf_gf_u8(p, q, o, 1, 4, 4, 3, 3, 5000, 1.0, 1, p, 1 << 20, None) == _ffi.RF_E_UNSUPPORTED
    # radii above 120 run the float kernels on float copies of the images: a larger workspace
    assert lib.rf_gf_u8(p, q, o, 1, 64, 64, 3, 3, 500, 1.0, 1, p, 1 << 16, None) == _ffi.RF_E_WORKSPACE
    assert lib.rf_gf_workspace_bytes(1, 64, 64, 3, 3, 500) > lib.rf_gf_workspace_bytes(1, 64, 64, 3, 3, 45)
    assert lib.rf_gf_u8(p, q, o, 1, 64, 64, 3, 3, 2, 1.0, 1, p, 16, None) == _ffi.RF_E_WORKSPACE
    assert lib.rf_cnn_reflectance_u8(p, None, None, 1, 4, 4, p, p, None) == _ffi.RF_E_BADARG
    assert lib.rf_cnn_reflectance_packed_u8(p, None, None, 1, 4, 4, p, p, None) == _ffi.RF_E_BADARG
    assert lib.rf_cnn_reflectance_packed_u8(p, q, None, 0, 4, 4, None, None, None) == _ffi.RF_OK
    assert lib.rf_cnn_pack_weights(p, None, None) == _ffi.RF_E_BADARG
    assert lib.rf_cnn_pack_weights(base, base + 64, None) == _ffi.RF_E_BADARG   # overlapping
    assert b"overlap" in lib.rf_last_error()
    # 256-byte header (per-image grey flags) + 12 float and 12 double planes
    assert lib.rf_gf_workspace_bytes(1, 100, 200, 3, 3, 45) == 256 + 100 * 200 * 12 * 12
    assert lib.rf_colorize_workspace_bytes(0) == 0 and lib.rf_colorize_workspace_bytes(3) > 0
    assert lib.rf_colorize_srgb_u8(p, q, o, None, 1, 4, 4, 0, 0, None, p, 1 << 20, None) == _ffi.RF_E_BADARG
    assert lib.rf_colorize_srgb_u8(p, q, o, None, 1, 4, 4, 48, 0, p, p, 1 << 20, None) == _ffi.RF_E_BADARG
    assert b"rank" in lib.rf_last_error()
    assert lib.rf_colorize_srgb_u8(p, q, o, None, 1, 4, 4, 47, 15, p, p, 8, None) == _ffi.RF_E_WORKSPACE
    assert lib.rf_gf_workspace_bytes(1, 100, 200, 3, 2, 45) == 0
    with pytest.raises(ValueError):
        _ffi.check(_ffi.RF_E_BADARG, "x")
    with pytest.raises(_ffi.RFError):
        _ffi.check(_ffi.RF_E_HIP, "x")


def test_operators_fail_loudly_without_gpu(built):
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    img = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(_ffi.RFError):
        rf.apply_filter("bilateral", img, img, 20, 22)
    with pytest.raises(_ffi.RFError):
        rf.apply_filter("guided", img, img, 3, 5)
    with pytest.raises(_ffi.RFError):
        rf.get_reflectance_caffe(rf.decompose_with_trained_CNN.ReflectanceNet(), img)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "librf_hip.so"))
    with pytest.raises(_ffi.RFError):
        _ffi.load_library()
