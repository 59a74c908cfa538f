"""Build-time guard for the LDS-DMA pipelines of csrc/gemm_fast.hip (CPU only: hipcc cross-compiles for gfx950).

The GEMM main loops keep the NEXT slabs' LDS-DMA pieces in flight while the current slab is read and multiplied; ordering is the
hand-counted `s_waitcnt vmcnt(N)` + `s_barrier` pairs in the source.  Twice this round the COMPILER put its own `s_waitcnt vmcnt(0)`
into that loop - once because `__builtin_amdgcn_global_load_lds` makes it assume the next `ds_read` may alias the pending LDS write,
once because its wait-count bookkeeping carried the epilogue's stores into the next tile's loop - and each time the kernels silently ran
at one DMA round trip per slab.  This test compiles the file to assembly and checks that, between the source's counted wait of a slab
and the end of that slab's 32 MFMAs, every `s_waitcnt vmcnt` comes from the source's inline asm (inside ;;#ASMSTART / ;;#ASMEND), and that
the slab's DMA pieces are issued between its MFMAs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tensorflowasr_amd", "csrc", "gemm_fast.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

KERNELS = [  # mangled-name fragments: forward FFN (bias + swish), its data gradient (swish' + dropout), plain NT product
    "gemm_fast_kernelILb0ELb0ELi128ELi5E",
    "gemm_fast_kernelILb0ELb1ELi128ELi6E",
    "gemm_fast_kernelILb0ELb1ELi128ELi0E",
]


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "gemm_fast.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "--cuda-device-only", "-S",
           SRC, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().split("\n")


def _body(lines, frag):
    start = next(i for i, l in enumerate(lines) if frag in l and l.rstrip().endswith(":") is False and re.match(r"^_ZN.*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end]


@pytest.mark.parametrize("frag", KERNELS)
def test_no_compiler_made_vmcnt_wait_inside_the_slab_loop(asm, frag):
    body = _body(asm, frag)
    in_asm, counted = False, None
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm and re.search(r"s_waitcnt vmcnt\((8|6)\)", l):
            counted = i
            break
    assert counted is not None, "the hand-counted wait of the slab loop was not found"
    mfma = pieces_between = 0
    in_asm = True  # we start inside the asm block that holds the counted wait
    seen_first_mfma = False
    for l in body[counted + 1:]:
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        if "v_mfma_f32_16x16x32" in l:
            mfma += 1
            seen_first_mfma = True
            if mfma == 32:
                break
        elif "global_load_lds_dwordx4" in l and seen_first_mfma:
            pieces_between += 1
        elif re.search(r"s_waitcnt\s+vmcnt", l) and not in_asm:
            raise AssertionError(f"compiler-made vector-memory wait inside the slab loop of {frag}: {l.strip()}")
    assert mfma == 32
    assert pieces_between >= 6, f"the next-next slab's DMA pieces are no longer issued between the MFMAs ({pieces_between})"


# ---------------------------------------------------------------------------------------------------------------------
# 256-row GEMM (csrc/gemm_big.h, the joint vocabulary projection = bench.py's `roofline` kernel): the epilogue reads its 128 pinned
# accumulators with explicit v_accvgpr_read.  Re-defining a fragment as an accumulation-register VALUE (asm "+a" on a copy) made the
# compiler build a register window: 28 v_accvgpr_mov + 4 v_accvgpr_write per fragment block; and fmaxf on a DPP-moved value costs three
# instructions per reduction step instead of one v_max_f32_dpp (common.h row16_max).
def test_gemm_big_epilogue_reads_accumulators_in_place(asm):
    body = _body(asm, "gemm_big_kernelILb0ELi256ELi64ELb0ELb0E")  # the row-oriented variant (TFASR_BIG_TR=0)
    last_mfma = max(i for i, l in enumerate(body) if "v_mfma_f32_16x16x32" in l)
    epi = body[last_mfma:]
    n_mov = sum("v_accvgpr_mov_b32" in l for l in epi)
    n_read = sum("v_accvgpr_read_b32" in l for l in epi)
    assert n_mov == 0, f"{n_mov} v_accvgpr_mov in the epilogue: the accumulators are being shuffled again"
    assert n_read == 128, f"{n_read} accumulator reads (one per element of the wave's 64 x 128 tile expected)"
    n_dpp_max = sum("v_max_f32_dpp" in l for l in epi)
    assert n_dpp_max == 64, f"{n_dpp_max} v_max_f32_dpp (16 row reductions x 4 steps expected)"
    # the canonicalising `v_max_f32 vN, vN, vN` of llvm.maxnum on a DPP-moved value must be gone
    canon = [l for l in epi if re.match(r"\s*v_max_f32_e32 (v\d+), (v\d+), (v\d+)\s*$", l) and len(set(re.findall(r"v\d+", l)[1:])) == 1]
    assert len(canon) <= 2, canon[:4]


def test_gemm_big_transposed_epilogue_stores_from_the_accumulators(asm):
    """Round 5: the joint projection's default kernel keeps its accumulators TRANSPOSED (operand slots of every MFMA swapped), so a lane owns
    four consecutive columns of one row: the epilogue reads the 128 accumulators in place, stores 8-byte row pieces straight from registers
    and never writes a transposition strip to LDS (only the tile's bias vector), and nothing is spilled."""
    body = _body(asm, "gemm_big_kernelILb0ELi256ELi64ELb0ELb1E")
    last_mfma = max(i for i, l in enumerate(body) if "v_mfma_f32_16x16x32" in l)
    epi = body[last_mfma:]
    assert sum("v_accvgpr_mov_b32" in l for l in epi) == 0
    assert sum("v_accvgpr_read_b32" in l for l in epi) == 128
    assert sum(bool(re.search(r"\bds_write", l)) for l in epi) <= 2, "an LDS strip is back in the transposed epilogue"
    assert sum("global_store_dwordx4" in l for l in epi) >= 16, "the 16-byte row pieces are no longer stored from registers"
    assert not any("scratch_" in l for l in body), "spill in the 256-row kernel (a scratch reload costs the DMA prefetch)"


# ---------------------------------------------------------------------------------------------------------------------
# log-mel front end (csrc/logmel.hip): the next frame's samples are prefetched at the top of the frame loop and consumed in front of
# the frame's stores.  Three compiler behaviours broke that silently while it was written (a select behind each load, a phantom
# fetch-without-prepare path, the consumption sunk behind the stores); each shows up as a vmcnt wait where none belongs.
LOGMEL_SRC = os.path.join(ROOT, "tensorflowasr_amd", "csrc", "logmel.hip")


@pytest.fixture(scope="module")
def logmel_asm(tmp_path_factory):
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_lm") / "logmel.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "--cuda-device-only", "-S",
           LOGMEL_SRC, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().split("\n")


@pytest.mark.parametrize("frag", ["logmel_kernelItEE", "logmel_kernelIfEE"])
def test_logmel_frame_loop_prefetch_is_not_drained(logmel_asm, frag):
    body = _body(logmel_asm, frag)
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    # the frame loop = the longest backward branch
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
            best = (labels[m.group(1)], i)
    assert best is not None
    loop = body[best[0]:best[1]]
    loads = [i for i, l in enumerate(loop) if re.match(r"\s+global_load_dword\s", l)]
    assert len(loads) >= 16, "the 16 sample loads of the next frame are expected inside the frame loop"
    first, last = loads[0], loads[15]
    # (1) nothing waits for vector memory between the loop top and the last of the 16 requests (they leave back to back)
    for l in loop[:last]:
        assert not re.search(r"s_waitcnt\s+vmcnt", l), f"vector-memory wait in front of / between the prefetch loads: {l.strip()}"
    # (2) the output stores are the untracked inline-asm ones, and the waits for the prefetched samples come BEFORE them
    stores = [i for i, l in enumerate(loop) if "global_store" in l]
    assert stores, "output stores not found in the frame loop"
    waits = [i for i, l in enumerate(loop) if re.search(r"s_waitcnt\s+vmcnt\(0\)", l)]
    assert waits and min(waits) < min(stores), "the prefetched samples must be consumed in front of the frame's stores"
    in_asm = False
    for i, l in enumerate(loop):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif "global_store" in l:
            assert in_asm, "a compiler-tracked store in the frame loop puts vmcnt(0) (= wait for the stores) in front of the next frame"
    # (3) no vector-memory wait after the stores inside the loop (the loop-carried samples were consumed already)
    tail = [l for l in loop[max(stores):] if re.search(r"s_waitcnt\s+vmcnt", l)]
    # the only loads behind the stores are the mel-band tails read from the dense matrix (bands wider than the LDS taps): their wait is inside that inner loop
    assert len(tail) <= 1, tail
