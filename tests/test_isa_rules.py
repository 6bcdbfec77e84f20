"""Static rules on the compiled kernels (hipcc cross-compiles here; no GPU needed)."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))


def _hipcc():
    for p in ("/opt/rocm/bin/hipcc",):
        if Path(p).exists():
            return p
    pytest.skip("hipcc not found")


@pytest.mark.parametrize("src", ["attention_train.hip"])
def test_lds_direct_tiles_are_handed_over_behind_a_vmcnt_wait(tmp_path, src):
    """the bf16 dK/dV attention kernel once read an LDS-direct (global_load_lds) tile with no `s_waitcnt vmcnt` between
    the load and the LDS read -- `__syncthreads()` does not promise one.  Rule, checked on the ISA hipcc emits for gfx950: in a kernel
    that uses LDS-direct loads, a loop header that runs into an s_barrier has `vmcnt(0)` in front of that barrier
    (common.h: lds_dma_barrier).  tools/isa_scan.py holds the scan; the pre-fix assembly failed it in ten kernels."""
    from isa_scan import lds_dma_handover_findings

    out = tmp_path / (src + ".s")
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT / 'include'}", f"-I{ROOT / 'everyvoice_amd' / 'csrc'}", "-S",
                        "--cuda-device-only", "-o", str(out), str(ROOT / "everyvoice_amd" / "csrc" / src)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    assert "global_load_lds" in text  # (the rule looks at something)
    bad = lds_dma_handover_findings(out)
    assert not bad, bad[:3]
