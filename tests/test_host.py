"""CPU-side checks of the drop-in boundary: C-ABI symbol export, module surface, error behaviour,
and that the product never routes through the oracle or any CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden

REF_KEYS = {  # state_dict of the reference module (functions.py:19-24) for in_dim = C
    "gamma": lambda C: (1,),
    "query_conv.weight": lambda C: (C // 8, C, 1, 1), "query_conv.bias": lambda C: (C // 8,),
    "key_conv.weight": lambda C: (C // 8, C, 1, 1), "key_conv.bias": lambda C: (C // 8,),
    "value_conv.weight": lambda C: (C, C, 1, 1), "value_conv.bias": lambda C: (C,),
}


@pytest.fixture(scope="module")
def device_lib_path():
    import __graft_entry__ as g
    g.build()                      # hipcc cross-compiles gfx950 without a GPU
    from ccnet_amd import _lib
    return _lib.LIB_PATH


def test_device_library_exports_every_declared_symbol(device_lib_path):
    from ccnet_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 15 and set(names) == set(_lib._PROTOTYPES)
    dll = ctypes.CDLL(device_lib_path)          # loads without a GPU; no compute call is made
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/ccnet_cca.h but not exported"
    # ... and NOTHING ELSE is a dynamic symbol (align256, find_word_option, softmax_backward_impl and the
    # kernels' device stubs used to sit next to the C ABI): -fvisibility=hidden + csrc/exports.map
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", device_lib_path], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(names), sorted(set(exported) ^ set(names))
    lib = _lib.CcaLibrary(device_lib_path)
    assert lib.ccnet_cca_version() == 220 and lib.ccnet_cca_arch() == b"gfx950"
    # argument validation happens before any launch, so it is checkable here
    assert lib.ccnet_ca_forward_f32(None, None, None, 1, 1, 2, 2, 0, None) == -2
    assert lib.ccnet_ca_forward_f32(None, None, None, 0, 1, 2, 2, 0, None) == -1
    assert "ccnet_cca" in lib.last_error()
    assert lib.ccnet_ca_softmax_backward_workspace_bytes(8, 97, 97) == ((8 * 97 * 97 + 3) // 4) * 4


def test_device_library_contains_gfx950_code_object(device_lib_path):
    blob = open(device_lib_path, "rb").read()
    assert b"gfx950" in blob and b"weight_strip_kernel" in blob and b"map_strip_kernel" in blob


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
# kernels that may still use scratch: none of them is on a default route of the headline / configs[4] steps (long-strip
# windowed kernels); everything else must be spill-free
SCRATCH_ALLOWED = ("map_long_kernel",)          # the windowed kernels of strips 101 .. 320 (cca_long.hpp): SGPR spills + <= 84 B of scratch
# (round 4 emptied the rest of the list: the packed split-bf16 dA strip kernel -- 126 spilled VGPRs -- and the split-bf16 dq | dk row
#  strip kernel -- 36 B of scratch -- left the library)


def code_object_kernels(lib_path, tmp_path):
    """{demangled-ish kernel name: metadata dict} read from the gfx950 code object inside the shipped library."""
    import subprocess
    fat, co = str(tmp_path / "cca.fatbin"), str(tmp_path / "cca.co")
    subprocess.run([f"{LLVM_BIN}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
    subprocess.run([f"{LLVM_BIN}/clang-offload-bundler", "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM_BIN}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "name" and val.startswith("_ZN3cca"):
            cur = kernels.setdefault(val, {})
        elif key == "name":
            cur = None if not val.startswith("_Z") else cur
        elif cur is not None and key in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count"):
            cur[key] = int(val)
    return kernels


@pytest.mark.skipif(not os.path.exists(f"{LLVM_BIN}/clang-offload-bundler"), reason="no LLVM binutils")
def test_hot_path_kernels_have_no_scratch_and_no_spilled_vgprs(device_lib_path, tmp_path):
    """0 spilled VGPRs and 0 bytes of scratch in every kernel a default route can reach (read from the
    code-object notes of the library that ships, not from a compiler remark)."""
    kernels = code_object_kernels(device_lib_path, tmp_path)
    assert len(kernels) > 20
    bad = {n: k for n, k in kernels.items()
           if (k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0))
           and not any(a in n for a in SCRATCH_ALLOWED)}
    assert not bad, bad


def test_integration_md_shows_the_tested_stub_verbatim():
    """INTEGRATION.md section 2 claims to show tests/reference_side_stub.py -- the reference-side ctypes binding that
    tests/test_gpu_parity.py::test_reference_side_stub_binds_the_fast_kernels runs on the GPU -- verbatim."""
    stub = open(os.path.join(ROOT, "tests", "reference_side_stub.py")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "```python\n" + stub + "```" in doc


def test_missing_library_fails_loudly(tmp_path):
    from ccnet_amd import _lib
    with pytest.raises(_lib.CcaError, match="no CPU or PyTorch fallback"):
        _lib.CcaLibrary(str(tmp_path / "libccnet_cca.so"))


@pytest.mark.parametrize("C", [16, 64, 512])
def test_module_surface_matches_reference(C):
    from cc_attention import CrissCrossAttention as ViaDropIn
    from cc_attention.functions import CrissCrossAttention as ViaFunctions
    from ccnet_amd import CrissCrossAttention
    assert ViaDropIn is CrissCrossAttention and ViaFunctions is CrissCrossAttention
    m = CrissCrossAttention(C)
    sd = m.state_dict()
    assert set(sd) == set(REF_KEYS)
    for k, shp in REF_KEYS.items():
        assert tuple(sd[k].shape) == shp(C), k
    assert float(m.gamma) == 0.0                                   # functions.py:24
    assert isinstance(m.softmax, torch.nn.Softmax) and m.softmax.dim == 3
    assert callable(m.INF)
    inf = m.INF(2, 3, 4)                                           # functions.py:11-12 semantics
    assert tuple(inf.shape) == (8, 3, 3)
    assert torch.isneginf(torch.diagonal(inf, dim1=1, dim2=2)).all()
    off = inf[~torch.eye(3, dtype=torch.bool).expand(8, 3, 3)]
    assert (off == 0).all()


def test_reference_checkpoint_loads_strictly():
    from ccnet_amd import CrissCrossAttention
    g = load_golden("small_2x64x8x8")
    m = CrissCrossAttention(64)
    res = m.load_state_dict({k[len("param."):]: v for k, v in g.items() if k.startswith("param.")}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert float(m.gamma) == 0.5


def test_cpu_input_raises_instead_of_falling_back():
    from ccnet_amd import CA_Map, CA_Weight, CrissCrossAttention, criss_cross_attention
    m = CrissCrossAttention(16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 16, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CA_Weight.apply(torch.randn(1, 2, 4, 4), torch.randn(1, 2, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CA_Map.apply(torch.randn(1, 4, 4, 8), torch.randn(1, 16, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        criss_cross_attention(torch.randn(1, 2, 4, 4), torch.randn(1, 2, 4, 4), torch.randn(1, 16, 4, 4),
                              torch.randn(1, 16, 4, 4), torch.zeros(1))


def test_product_never_imports_the_oracle_or_the_emulator():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    bad = []
    for pkg in ("ccnet_amd", "cc_attention"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h")):
                    text = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "cca_oracle" in text:
                        bad.append(os.path.join(dirpath, f))
                    if f.endswith(".py") and ("hip_emu" in text or "libcca_emu" in text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", bench)]
    lo = bench.index("def cpu_baseline")
    hi = bench.index("\ndef ", lo + 1)
    assert len(uses) == 1 and lo < uses[0] < hi, "bench.py may use the oracle only inside cpu_baseline()"


def test_bench_accounting_matches_survey():
    import bench
    assert bench.core_bytes(8, 512, 97, 97) == 1_040_560_128
    assert round(bench.core_flops(8, 512, 97, 97) / 1e9, 2) == 50.47
    nbytes, flops = bench.kernel_accounting("weight", 8, 512, 97, 97, row=False)
    assert nbytes == 2 * 4 * 8 * 512 * 97 * 97 + 4 * 8 * 97 * 97 * 97
    assert flops == 2 * 8 * 97 * 97 * 97 * 512


def test_fusable_rejects_modified_projections():
    """the one-node path bypasses nn.Conv2d.forward, so anything but the constructor's plain dense 1x1
    convolutions (geometry, dtype, device, hooks) must fall back to torch's own conv2d."""
    import torch
    from ccnet_amd import CrissCrossAttention
    m = CrissCrossAttention(16)
    x = torch.zeros(1, 16, 3, 3)
    assert m._fusable() and m._fusable(x)
    assert not m._fusable(x.double())
    m.key_conv.stride = (2, 2)
    assert not m._fusable()
    m.key_conv.stride = (1, 1)
    m.value_conv.half()
    assert not m._fusable() and not m._fusable(x)
    m.value_conv.float()
    h = m.query_conv.register_full_backward_pre_hook(lambda mod, g: None)
    assert not m._fusable()
    h.remove()
    assert m._fusable(x)


def test_inf_matches_reference_values():
    import torch
    from ccnet_amd.functions import INF
    t = INF(2, 3, 4, device="cpu")
    assert t.shape == (2 * 4, 3, 3) and torch.isneginf(t.diagonal(dim1=1, dim2=2)).all()
    assert (t[~torch.isinf(t)] == 0).all()


def test_product_sources_carry_no_emulator_code_and_no_env_knobs():
    """the emulator primitives live under tests/emu/ (same header name, earlier on the include
    path), and the product path is not steered by environment variables."""
    csrc = os.path.join(ROOT, "ccnet_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            text = open(os.path.join(csrc, f)).read()
            assert "CCNET_EMU" not in text and "hip_emu" not in text, f
            assert "getenv" not in text, f
    assert os.path.exists(os.path.join(ROOT, "tests", "emu", "cca_platform.hpp"))


def test_measured_traffic_is_keyed_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py quotes PMC traffic only for a build of the kernel sources it was measured on."""
    import json
    import bench
    from ccnet_amd import _lib
    sha = _lib.kernel_source_sha16()
    assert len(sha) == 16 and sha == _lib.kernel_source_sha16()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    good = {"_step_total_bytes": 123, "_src_sha16": sha}
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(good))
    assert bench.measured_traffic(None) == good
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps({"_step_total_bytes": 123, "_src_sha16": "0" * 16}))
    assert bench.measured_traffic(None) is None
    assert bench.measured_traffic(None, "missing.json") is None


def test_options_and_workspace_sizes_are_host_logic_behind_two_entry_points(device_lib_path):
    """The process-wide options live behind ccnet_cca_set_option / _get_option by name and every workspace size behind
    ccnet_cca_workspace_bytes(entry, ...) (29 exported symbols).  Pure host logic: no kernel is launched."""
    from ccnet_amd import _lib
    lib = _lib.get_lib()
    assert len(_lib.declared_symbols()) <= 39          # (30 + the weight packer, the split-with-column-sums producer and three device-state probes of round 5 + the three-plane backward and the three projection GEMMs of round 6)
    for name, default in (("impl", _lib.CCNET_IMPL_AUTO), ("precision", _lib.CCNET_PRECISION_DEFAULT), ("branch_mask", 3),
                          ("planes_ring", 2), ("planes_stream", 1), ("planes_overlap", -1), ("planes_xcd", 1), ("energy_tail", 1), ("da_stages", 2), ("dqdk_wpc3", 1), ("dqdk_exact", 1), ("bf16_partial", 1)):
        assert lib.get_option(name) == default, name
    assert lib.set_option("planes_overlap", 0) == -1 and lib.get_option("planes_overlap") == 0
    assert lib.set_option("planes_overlap", -1) == 0
    # status and value never share an int -- -1 is a legal value of "planes_overlap", errors come back as codes
    import ctypes
    prev, val = ctypes.c_int(7), ctypes.c_int(7)
    assert lib.ccnet_cca_set_option(b"impl", 99, ctypes.byref(prev)) == -3 and prev.value == 7             # invalid: unchanged
    assert lib.get_option("impl") == _lib.CCNET_IMPL_AUTO
    assert lib.ccnet_cca_set_option(b"planes_overlap", 3, None) == -3 and lib.ccnet_cca_set_option(b"planes_ring", -1, None) == -3
    assert lib.ccnet_cca_set_option(b"no_such_option", 1, None) == -3 and lib.ccnet_cca_get_option(b"no_such_option", ctypes.byref(val)) == -3
    assert b"unknown option" in lib.ccnet_cca_last_error_string() and val.value == 7
    assert lib.ccnet_cca_get_option(b"planes_overlap", ctypes.byref(val)) == 0 and val.value == -1
    with pytest.raises(_lib.CcaError):
        lib.set_option("precision", 17)
    assert lib.ccnet_cca_version() == _lib.CCNET_CCA_VERSION == 220
    B, C, Cq, H, W = 8, 512, 64, 97, 97
    px = B * H * W * 4
    sm = lib.ccnet_ca_softmax_backward_workspace_bytes(B, H, W)
    assert sm > 0 and lib.ccnet_cca_workspace_bytes(_lib.CCNET_WS_SOFTMAX_BACKWARD, B, 0, 0, H, W) == sm
    assert lib.ccnet_cca_pm_workspace_bytes(B, C, Cq, H, W, 0) == px * C                       # the column partial
    pmb = lib.ccnet_cca_pm_workspace_bytes(B, C, Cq, H, W, 1)
    assert pmb >= sm + px * C + px * 2 * Cq                                                     # + softmax slabs + the dq | dk partials
    plb = lib.ccnet_cca_planes_workspace_bytes(B, C, Cq, H, W, 1)
    assert plb >= pmb + B * H * W * 2 * C * 2                                                   # + dy as planes
    assert lib.ccnet_cca_planes_workspace_bytes(B, C, Cq, H, W, 0) == px * C
    assert lib.ccnet_cca_workspace_bytes(99, B, C, Cq, H, W) == 0 and lib.ccnet_cca_workspace_bytes(_lib.CCNET_WS_PM_FORWARD, 0, C, Cq, H, W) == 0


def test_module_routing_table(device_lib_path):
    """``CrissCrossAttention.forward`` is ONE routing table.  ``route`` is a pure function of the input's
    dtype / layout / shape, the module's flags and the two process-wide knobs; enumerate it (no kernel is launched: the
    tensors are meta-free CPU stand-ins carrying only dtype, strides and shape)."""
    from ccnet_amd import CrissCrossAttention, _lib
    lib = _lib.get_lib()
    m = CrissCrossAttention(64)
    nchw = lambda B, H, W, dt=torch.float32: torch.empty(B, 64, H, W, dtype=dt)                       # noqa: E731
    cl = lambda B, H, W, dt=torch.float32: nchw(B, H, W, dt).contiguous(memory_format=torch.channels_last)   # noqa: E731
    assert set(m.ROUTES) >= {m.route(nchw(1, 8, 8))} and len(m.ROUTES) <= 5          # 
    table = {
        ("f32 NCHW 97x97 B=8", lambda: m.route(nchw(8, 97, 97))): "f32-planes",
        ("f32 NCHW 97x97 B=1", lambda: m.route(nchw(1, 97, 97))): "f32-planes",
        ("f32 NCHW 129x129 (plane kernels padded to 132 positions)", lambda: m.route(nchw(2, 129, 129))): "f32-planes",
        ("f32 NCHW 129x257 (rows in blocks of <= 132 positions)", lambda: m.route(nchw(2, 129, 257))): "f32-planes",
        ("f32 NCHW 97x193", lambda: m.route(nchw(1, 97, 193))): "f32-planes",
        ("f32 NCHW 161x321 (both sides beyond 132: column AND row passes in blocks)", lambda: m.route(nchw(1, 161, 321))): "f32-planes",
        ("f32 NCHW 257x513 (evaluate.py:146-166 at scale 2)", lambda: m.route(nchw(1, 257, 513))): "f32-planes",
        ("f32 NCHW 600x140 (columns beyond 4 blocks)", lambda: m.route(nchw(1, 600, 140))): "separate-strips",
        ("f32 NCHW 257x129 (tall: column passes in blocks)", lambda: m.route(nchw(1, 257, 129))): "f32-planes",
        ("f32 NCHW 129x600 (rows beyond 4 blocks)", lambda: m.route(nchw(1, 129, 600))): "separate-strips",
        ("f32 channels_last (one copy to NCHW, the output back in channels_last)", lambda: m.route(cl(2, 33, 18))): "f32-planes",
    }
    for (what, fn), want in table.items():
        assert fn() == want, what
    with torch.no_grad():       # inference (evaluate.py:246) takes the same routes
        assert m.route(nchw(1, 129, 257)) == "f32-planes" and m.route(nchw(1, 161, 321)) == "f32-planes"
    m.to(torch.bfloat16)
    assert m.route(cl(2, 129, 129, torch.bfloat16)) == "bf16-pixel-major"
    assert m.route(nchw(1, 129, 257, torch.bfloat16)) == "f32-planes-cast"      # beyond the bf16 kernels: blocked fp32 plane kernels on copies
    assert m.route(nchw(1, 257, 513, torch.bfloat16)) == "f32-planes-cast"
    assert m.route(nchw(1, 330, 9, torch.bfloat16)) == "f32-planes-cast"         # (round 3: any-shape fp32 kernels through fp32 copies)
    assert m.route(nchw(1, 200, 9, torch.bfloat16)) == "f32-planes-cast"         # (round 3: windowed fp32 kernels through fp32 copies)
    assert m.route(nchw(1, 600, 9, torch.bfloat16)) == "separate-strips"           # beyond 528 positions: any-shape fp32 kernels through fp32 copies
    m.to(torch.float16)
    assert m.route(nchw(2, 97, 97, torch.float16)) == "f32-planes-cast"          # fp16: no native kernels, the fp32 node on copies
    m.to(torch.float32)
    m.to(torch.float32)
    m.split_planes = False
    assert m.route(nchw(2, 97, 97)) == "separate-strips" and m.route(nchw(8, 97, 97)) == "separate-strips"
    m.split_planes = True
    m.recompute_attention = True                     # (the fast routes honour the flag themselves now)
    assert m.route(nchw(2, 97, 97)) == "f32-planes" and m.route(cl(2, 33, 18)) == "f32-planes"
    m.recompute_attention = False
    m.fuse_projections = False
    assert m.route(nchw(2, 97, 97)) == "separate-strips"
    m.fuse_projections = True
    prev = lib.ccnet_cca_set_precision(_lib.CCNET_PRECISION_F32)                 # (the knobs are honoured)
    try:
        assert lib.ccnet_cca_get_precision() == _lib.CCNET_PRECISION_F32
        assert m.route(nchw(2, 97, 97)) == "separate-strips" and m.route(cl(2, 33, 18)) == "separate-strips"
    finally:
        lib.ccnet_cca_set_precision(prev)
    previ = lib.ccnet_cca_set_impl(_lib.CCNET_IMPL_DIRECT)
    try:
        assert m.route(nchw(2, 97, 97)) == "separate-strips"
    finally:
        lib.ccnet_cca_set_impl(previ)
    assert m.route(nchw(2, 97, 97)) == "f32-planes"
    m.query_conv = torch.nn.Conv2d(64, 8, 3, padding=1)                           # a swapped-out projection: nothing is bypassed
    assert m.route(nchw(2, 97, 97)) == "separate-strips"
