"""The register audit (tools/check_owned_regs.py) on synthetic assembly: it must look at the whole function, also
behind an early exit, and tell the inline asm's own accumulator accesses from the compiler's."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_owned_regs", os.path.join(ROOT, "tools", "check_owned_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


KERNEL = "_ZN2fa4demoEv"
HEAD = KERNEL + ":\n\ts_load_dwordx2 s[0:1], s[0:1], 0x0\n\ts_cbranch_scc1 .LBB0_2\n\ts_endpgm\n.LBB0_2:\n"
OWNED = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]\n\t;;#ASMEND\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n\t.amdhsa_kernel " + KERNEL + "\n\t\t.amdhsa_private_segment_fixed_size 0\n\t.end_amdhsa_kernel\n"


def test_audit_accepts_asm_owned_accumulators(capsys):
    _tool().audit(HEAD + OWNED + "\tv_add_f32_e32 v0, v1, v2\n" + TAIL, KERNEL)


@pytest.mark.parametrize("line", ["v_accvgpr_write_b32 a4, v8", "v_accvgpr_read_b32 v9, a1", "scratch_store_dword off, v9, off",
                                  "ds_read_b64_tr_b16 a[0:1], v113 offset:40960"])
def test_audit_sees_compiler_use_behind_an_early_exit(line):
    with pytest.raises(SystemExit):
        _tool().audit(HEAD + OWNED + "\t" + line + "\n" + TAIL, KERNEL)
