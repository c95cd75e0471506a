"""CPU: libserhip.so builds for gfx950, loads, and exports exactly the C ABI of include/ser_hip.h."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ser_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ser_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(built_library)
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ser_hip.h but not exported"


def test_binding_covers_header(built_library):
    from interspeech_ser_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()
    assert _lib.lib.ser_version() == _lib.ABI_VERSION == int(re.search(r"#define SER_ABI_VERSION (\d+)", open(HEADER).read()).group(1))


def test_mode_and_op_constants_match_header(built_library):
    """SER_MODE_* / SER_ACT_* / SER_OP_* / SER_WS_* of the header are the numbers the ctypes binding passes."""
    from interspeech_ser_amd import _lib
    text = open(HEADER).read()
    defs = {k: int(v) for k, v in re.findall(r"#define (SER_(?:MODE|ACT|OP|WS)_\w+)\s+(\d+)", text)}
    assert defs["SER_MODE_BF16"] == _lib.MODE_BF16 and defs["SER_MODE_FP32X"] == _lib.MODE_FP32X and defs["SER_MODE_FP16"] == _lib.MODE_FP16
    assert defs["SER_MODE_FP16X"] == _lib.MODE_FP16X and defs["SER_MODE_FP16Q"] == _lib.MODE_FP16Q
    assert defs["SER_ACT_NONE"] == _lib.ACT_NONE and defs["SER_ACT_GELU"] == _lib.ACT_GELU
    assert defs["SER_WS_LOGMEL"] == _lib.WS_LOGMEL and defs["SER_WS_WAVE_FRAMES"] == _lib.WS_WAVE_FRAMES
    ops = ("GEMM", "ATTENTION", "LAYERNORM", "WAVE_FRAMES", "ROW_CENTER", "LOGMEL", "PACK_ACT")
    assert [defs["SER_OP_" + o] for o in ops] == [getattr(_lib, "OP_" + o) for o in ops]


def test_struct_layouts_match_c(built_library, tmp_path):
    """Every ctypes mirror (ser_gemm_args, the command-list argument structs, ser_cmd) must have the C
    compiler's size and field offsets."""
    from interspeech_ser_amd._lib import STRUCT_MIRRORS
    lines = []
    for cname, cls in STRUCT_MIRRORS.items():
        lines.append(f'printf("{cname}.sizeof %zu\\n", sizeof({cname}));')
        for f, *_ in cls._fields_:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    src = tmp_path / "layout.c"
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\n'
                   f'int main(void){{' + "".join(lines) + 'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in STRUCT_MIRRORS.items():
        assert int(out[f"{cname}.sizeof"]) == ctypes.sizeof(cls), cname
        for f, *_ in cls._fields_:
            assert int(out[f"{cname}.{f}"]) == getattr(cls, f).offset, (cname, f)


def test_argument_errors_are_reported_not_raised(built_library):
    """Launchers return <0 and set ser_last_error() on bad arguments (no GPU needed: validation
    happens before any launch)."""
    from interspeech_ser_amd import _lib
    g = _lib.GemmArgs()
    assert _lib.lib.ser_gemm(ctypes.byref(g), None) < 0
    assert b"ser_gemm" in _lib.lib.ser_last_error()
    assert _lib.lib.ser_layernorm(None, 0, None, None, 1e-5, 0, None, 0, None, 0, 0, 1, 1, 8, None) < 0
    assert _lib.lib.ser_attention(None, 0, 0, 0, 0, 0, None, 1, 1, None, 0, None, None, 0, 0, 1, 64, 0.125, 1, 0, None, None, None, 0, None) < 0
    assert _lib.lib.ser_workspace_bytes(_lib.WS_LOGMEL, 4, 0, 0, 0, 1) == 4 * 1024 + 400 * 208 * 16 + 400 * 4


def test_engine_refuses_to_run_without_a_gpu(built_library):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd._lib import SerHipError
    from interspeech_ser_amd.engine import SpeechEncoder
    with pytest.raises(SerHipError):
        SpeechEncoder(C.TINY_WAVLM, {}, "cuda:0")


def test_command_list_reports_the_failing_entry(built_library):
    """ser_run validates like the launchers it wraps: a bad entry stops the list and names its index (no GPU needed)."""
    from interspeech_ser_amd import _lib
    cmds = (_lib.Cmd * 3)()
    cmds[0].op = _lib.OP_GEMM                       # null operands -> ser_gemm refuses before any launch
    failed = ctypes.c_int32(-1)
    assert _lib.lib.ser_run(cmds, 1, ctypes.byref(failed), None) < 0 and failed.value == 0
    assert b"ser_gemm" in _lib.lib.ser_last_error()
    cmds[0].op = 99
    assert _lib.lib.ser_run(cmds, 1, ctypes.byref(failed), None) < 0
    assert b"unknown op" in _lib.lib.ser_last_error()
    assert _lib.lib.ser_run(cmds, 0, None, None) == 0


def test_deberta_launchers_validate_arguments(built_library):
    from interspeech_ser_amd import _lib
    assert _lib.lib.ser_embed_ln_masked(None, None, None, None, 1e-7, None, None, None, 0, 1, 1, 8, 64, None) < 0
    assert b"ser_embed_ln_masked" in _lib.lib.ser_last_error()
    assert _lib.lib.ser_deberta_attention(None, 0, 0, 0, 0, 0, None, None, 0, 0, None, None, None, None, 0, 0, 1, 8, 1, 64, 1, None) < 0
    assert b"ser_deberta_attention" in _lib.lib.ser_last_error()


def test_deberta_bucket_map_of_the_host_equals_the_oracle(built_library):
    """The product's host-side log-bucket map (engine) and the oracle's are written separately; they must agree on every
    distance a 128-token sequence can produce, for the v3 setting (256 buckets, 512 positions) and the fixture's."""
    import torch
    from interspeech_ser_amd.engine import _deberta_log_bucket
    from oracle import ssl_oracle as O
    d = torch.arange(-511, 512)
    for buckets, maxpos in ((256, 512), (16, 512), (32, 128)):
        assert torch.equal(_deberta_log_bucket(d, buckets, maxpos), O.deberta_log_bucket(d, buckets, maxpos))
