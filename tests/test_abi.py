"""CPU checks of the C-ABI library: it is built, loads, and exports every symbol that
include/gpet_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpet_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    names = _declared_symbols()
    for must in ["gpet_ctx_create", "gpet_grad_image", "gpet_batch_create", "gpet_gp_fit_predict", "gpet_gp_factor",
                 "gpet_gp_normals", "gpet_gp_sample", "gpet_score_curves", "gpet_select_pixels", "gpet_trace_iterate",
                 "gpet_lml_batch", "gpet_last_error"]:
        assert must in names


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from gaussian_process_edge_trace_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table binds exactly the declared surface
    assert sorted(_lib.SYMBOLS) == _declared_symbols()
    assert _lib.load().gpet_abi_version() == 1


def test_struct_layouts_match_the_header(tmp_path):
    """ctypes mirrors vs what a C compiler makes of include/gpet_hip.h (the header is plain C)."""
    import subprocess
    from gaussian_process_edge_trace_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gpet_hip.h"\n'
                    'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gpet_params), '
                    'offsetof(gpet_params, nu), offsetof(gpet_params, score_thresh), offsetof(gpet_params, jitter), '
                    'sizeof(gpet_scalars), offsetof(gpet_scalars, n), offsetof(gpet_scalars, done));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    P, S = _lib.GpetParams, _lib.GpetScalars
    assert got == [ctypes.sizeof(P), P.nu.offset, P.score_thresh.offset, P.jitter.offset, ctypes.sizeof(S),
                   S.n.offset, S.done.offset]


def test_no_gpu_fails_loudly_without_fallback():
    """Without a HIP device the product refuses to run (no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gaussian_process_edge_trace_amd import _lib
    import numpy as np
    import gaussian_process_edge_trace_amd as pkg
    with pytest.raises(_lib.GpetError) as ei:
        _lib.Context(0)
    assert ei.value.code == _lib.ERR_NO_DEVICE
    with pytest.raises(_lib.GpetError):
        pkg.GP_Edge_Tracing(np.array([[0, 5], [15, 5]]), np.zeros((16, 16), np.float32))
    with pytest.raises(_lib.GpetError):
        pkg.gpet_utils.comp_grad_img(np.zeros((8, 8)), np.ones((3, 3)))


def test_missing_rccl_is_reported_not_crashed():
    """A host without RCCL: gpet_comm_unique_id returns GPET_ERR_UNSUPPORTED (round 5 called dlerror() twice and handed
    std::string a null pointer).  In a child process: the binding is cached per process."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from gaussian_process_edge_trace_amd import _lib; "
            "lib = _lib.load(); buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES); "
            "print('rc', lib.gpet_comm_unique_id(buf))" % ROOT)
    env = dict(os.environ, GPET_RCCL_LIB="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "rc 6" in out.stdout, out.stdout


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "gaussian_process_edge_trace_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f


def test_option_registry_round_trip_and_documented():
    """The process-wide tuning switches are ONE table behind gpet_set_option / gpet_get_option / gpet_option_info (host
    code: no GPU needed): enumerate, set / get, clamping, "automatic" reported as the largest value + 1 by the C entry
    point, unknown names rejected -- and every option has a row in INTEGRATION.md section 3b."""
    import os
    from gaussian_process_edge_trace_amd import _lib as L
    opts = L.options()
    assert len(opts) >= 20 and all(o["lo"] <= o["default"] <= o["hi"] and o["doc"] for o in opts.values())
    assert "rng4" in opts and "oj_persist" in opts and "struct_path" in opts
    old = L.set_option("oj_max_sweeps", 1000)
    try:
        assert L.get_option("oj_max_sweeps") == opts["oj_max_sweeps"]["hi"]  # clamped
        assert L.set_option("oj_max_sweeps", 7) == opts["oj_max_sweeps"]["hi"] and L.get_option("oj_max_sweeps") == 7
    finally:
        L.set_option("oj_max_sweeps", old)
    lib = L.load()
    o4 = L.get_option("rng4")
    try:
        L.set_option("rng4", -1)
        assert lib.gpet_set_option(b"rng4", 1) == opts["rng4"]["hi"] + 1  # -1 ("automatic") as hi + 1
        assert lib.gpet_set_option(b"rng4", 0) == 1
    finally:
        L.set_option("rng4", o4)
    assert lib.gpet_set_option(b"no_such_option", 1) == -1
    with pytest.raises(ValueError):
        L.get_option("no_such_option")
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in opts:
        assert "| `%s` |" % name in doc, name


def _device_disassembly(tmp_path):
    """gfx950 disassembly of every code object in the shipped library (llvm-objdump --offloading writes the bundles next to
    its input, so it works on a copy)."""
    import shutil
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    so = tmp_path / "lib.so"
    shutil.copy(ge.LIB, so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    text = []
    for f in sorted(os.listdir(tmp_path)):
        if "amdgcn" in f:
            text.append(subprocess.run([objdump, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


def _functions(dis, name_part):
    out, cur, name = {}, None, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            name = m.group(1)
            cur = out.setdefault(name, []) if name_part in name else None
            continue
        if cur is not None and line.startswith("\t"):
            cur.append(re.sub(r"\s+", " ", line.strip().split("//")[0].strip()))
    return out


def test_isa_every_wave_drains_its_row_stores_before_a_pair_slot_is_published(tmp_path):
    """k_oj_persist (csrc/gpet_eig.hip): a pair slot is counted finished by thread 0 after a workgroup barrier; the rows the
    other three waves wrote (write-through stores, another workgroup on another XCD reads them next) must have COMPLETED by
    then.  A workgroup barrier orders nothing in global memory and the compiler puts no wait in front of it on its own
    (round 4's shipped object had none -- advisor finding): every wave executes s_waitcnt vmcnt(0) between its last row
    store and the barrier.  Checked in the ISA of both template variants."""
    fns = _functions(_device_disassembly(tmp_path), "k_oj_persist")
    assert len(fns) == 2, list(fns)
    for name, ins in fns.items():
        stores = [i for i, s in enumerate(ins) if re.match(r"(global_store_dwordx2 \S+ a\[\d+:\d+\], off|flat_store_dwordx2 \S+ a\[\d+:\d+\]) sc1$", s)]
        assert len(stores) >= 16, (name, len(stores))  # the row stores of the update (16 per wave and column batch)
        last = stores[-1]
        bar = next(i for i in range(last, len(ins)) if ins[i].startswith("s_barrier"))
        between = ins[last + 1:bar]
        assert any(s.startswith("s_waitcnt") and "vmcnt(0)" in s for s in between), (name, between)
        assert not any(s.startswith(("global_store", "flat_store", "buffer_store")) for s in between), (name, between)
