"""Build-time checks on the gfx950 ISA of the kernels that issue LDS-DMA from inline asm.

`fa_glds16` / `fa_dma16` (llama_attn.hip), `ab_glds16` (llama_attn_bwd.hip) and `tk_glds16` / `tk_glds4`
(lru_topk_bf16.hip) write M0 inside an asm block that does not (cannot: M0 is a reserved register for hipcc's inline
asm) declare the clobber, and they are invisible to hipcc's vmcnt tracking. That is correct as long as
  (1) no compiler-emitted instruction of those kernels depends on M0 -- every M0 write is the asm block's own
      `s_mov_b32 m0, sN`, followed (after its s_nop) by the LDS-DMA instruction it serves, and nothing else reads M0
      (no s_movrel / v_movrel / s_set_gpr_idx, no LDS-DMA issued through the builtin in the same kernel);
  (2) every LDS-DMA is waited for by hand before the data is read by another wave: walking back from any s_barrier of
      such a kernel an `s_waitcnt vmcnt(N)` is met before any LDS-DMA instruction.
A violation would be silent data corruption, not a build error; this test disassembles what the Makefile builds and
checks both statements per kernel. Runs without a GPU (hipcc cross-compiles)."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "llamarec_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function"]
FILES = {"llama_attn.hip": [], "llama_attn_bwd.hip": [], "lru_topk_bf16.hip": ["-fno-honor-nans"]}

DMA = re.compile(r"^\s*(global_load_lds_\w+|buffer_load_\w+ .*\blds\b)")
M0_USERS = re.compile(r"^\s*(s_movrel\w*|v_movrel\w*|s_set_gpr_idx\w*|ds_gws_\w+|s_sendmsg\w*)\b")


def _kernels(asm_text):
    """{kernel symbol: [instruction lines]} for every .amdhsa_kernel of the module."""
    names = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm_text, flags=re.M)
    out = {}
    for n in names:
        m = re.search(r"^%s:[^\n]*\n(.*?)^\s*s_endpgm" % re.escape(n), asm_text, flags=re.M | re.S)
        assert m, n
        body = []
        for ln in m.group(1).splitlines():
            ln = ln.split(";")[0].rstrip()
            if ln.strip() and not ln.strip().startswith(".") and not ln.strip().endswith(":"):
                body.append(ln.strip())
        out[n] = body
    return out


@pytest.mark.parametrize("src", sorted(FILES))
def test_inline_asm_lds_dma_kernels_keep_m0_and_vmcnt_by_hand(src, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC] + FLAGS + FILES[src] + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)],
                   check=True, capture_output=True, cwd=CSRC)
    kernels = _kernels(out.read_text())
    assert kernels, src
    checked = 0
    for name, body in kernels.items():
        dma_at = [i for i, ln in enumerate(body) if DMA.match(ln)]
        if not dma_at:
            # a kernel without LDS-DMA must not touch M0 at all (nothing of ours needs it)
            assert not any(re.search(r"\bm0\b", ln) for ln in body), (src, name)
            continue
        checked += 1
        # (1) every instruction that names M0 is `s_mov_b32 m0, sN` and serves an LDS-DMA within the next 3 instructions
        for i, ln in enumerate(body):
            assert not M0_USERS.match(ln), (src, name, ln)
            if re.search(r"\bm0\b", ln):
                assert re.match(r"s_mov_b32 m0, s\d+$", ln), (src, name, ln)
                nxt = body[i + 1:i + 4]
                assert any(DMA.match(x) for x in nxt), (src, name, ln, nxt)
                assert all(DMA.match(x) or x.startswith("s_nop") for x in nxt[:[bool(DMA.match(x)) for x in nxt].index(True) + 1]), \
                    (src, name, nxt)
        # ... and every LDS-DMA has its own M0 write right in front of it (the asm block's, not a hoisted one)
        for i in dma_at:
            prev = body[max(0, i - 3):i]
            assert any(re.match(r"s_mov_b32 m0, s\d+$", x) for x in prev), (src, name, body[i], prev)
        # (2) hipcc does not know these DMAs, so none of ITS waits is there for them: walking back from every s_barrier
        # of such a kernel, an `s_waitcnt ... vmcnt(N)` must come before any LDS-DMA does (text order: inside a loop body
        # the stage's wait stands in front of the barrier and the next stage's DMAs behind it)
        for i, ln in enumerate(body):
            if ln.startswith("s_barrier"):
                for x in reversed(body[:i]):
                    if re.match(r"s_waitcnt\b.*vmcnt\(\d+\)", x):
                        break
                    assert not DMA.match(x), (src, name, "an LDS-DMA reaches an s_barrier without a vmcnt wait in between", x)
    assert checked >= 1, src


TRAIN_FILES = {"lru_train_scores.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "lru_train_blocks.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


@pytest.mark.parametrize("src", sorted(TRAIN_FILES))
def test_training_panel_kernels_keep_their_operands_in_registers(src, tmp_path):
    """The training step's panel kernels hold operand sets one step ahead of their MFMAs in registers. Twice this round hipcc put
    such a set on the STACK instead (a struct of float4 passed by reference to the load / compute lambdas; an array written with
    separate statements and read in a loop): scratch_store / scratch_load round trips in front of every MFMA block, no build
    error, 25-50 % slower. Every kernel of these files must have an empty private segment and no spilled register."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC] + FLAGS + TRAIN_FILES[src] + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)],
                   check=True, capture_output=True, cwd=CSRC)
    text = out.read_text()
    names = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, flags=re.M)
    assert len(names) >= 4, (src, names)
    for n in names:
        blk = re.search(r"^\s*\.amdhsa_kernel\s+%s\n(.*?)\.end_amdhsa_kernel" % re.escape(n), text, flags=re.M | re.S).group(1)
        assert re.search(r"\.amdhsa_private_segment_fixed_size\s+0\b", blk), (src, n, "uses scratch")
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)] + [int(x) for x in re.findall(r"\.sgpr_spill_count:\s+(\d+)", text)]
    assert spills and max(spills) == 0, (src, spills)
    assert "scratch_" not in text, src


def test_diag_prototypes_still_compile(tmp_path):
    """tools/diag/*.hip are stand-alone measurement programs (the GEMM K-loop prototypes, the MFMA issue sweep, the store-rate
    probe): they are evidence behind DESIGN.md's numbers and must keep building for gfx950."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    diag = os.path.join(REPO, "tools", "diag")
    srcs = sorted(f for f in os.listdir(diag) if f.endswith(".hip"))
    assert {"gemm4w.hip", "gemm_bm.hip", "mfma_f32_issue.hip", "store_rate.hip"} <= set(srcs)
    for f in srcs:
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-c", "--cuda-device-only", os.path.join(diag, f), "-o", str(tmp_path / (f + ".o"))],
                       check=True, capture_output=True)


def test_attn256_accumulator_file_is_left_alone(tmp_path):
    """llama_attn256.hip keeps O, Q and K in a[0:255] under names hipcc never sees; every asm statement clobbers the whole
    accumulator file so that the compiler cannot park a value of its own there (it did, in the first build: Q fragments
    overwritten). Checked on what the Makefile builds: no v_accvgpr_* outside the asm statements anywhere in the kernel, no
    scratch access in any basic block that holds MFMAs of the steady state (64 per block), every s_barrier behind a
    vector-memory wait, and every LDS-DMA with its own M0 write."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path / "llama_attn256.s"
    subprocess.run([HIPCC] + FLAGS + ["-fno-slp-vectorize", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-Wno-unused-variable", "-Wno-unused-value",
                                      "-S", "--cuda-device-only", os.path.join(CSRC, "llama_attn256.hip"), "-o", str(out)],
                   check=True, capture_output=True, cwd=CSRC)
    text = out.read_text()
    m = re.search(r"^(_Z\w*attn_mfma256_kernel\w*):[^\n]*\n(.*?)^\s*s_endpgm", text, flags=re.M | re.S)
    assert m, "attn_mfma256_kernel not found"
    in_asm, outside_acc, blocks, cur = False, [], [], []
    for ln in m.group(2).splitlines():
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        code = t.split(";")[0].strip()
        if not code:
            continue
        if re.match(r"^\.?\w+:$", code) and not in_asm:      # a compiler basic-block label
            blocks.append(cur)
            cur = []
            continue
        if not in_asm and code.startswith("v_accvgpr"):
            outside_acc.append(code)
        cur.append(code)
    blocks.append(cur)
    assert not outside_acc, outside_acc[:5]
    steady = [b for b in blocks if sum(x.startswith("v_mfma") for x in b) == 64]
    assert steady, "no 64-MFMA block found"
    for b in steady:
        assert not any(x.startswith("scratch_") for x in b), "scratch access in a steady-state key block"
    body = [x for b in blocks for x in b]
    for i, ln in enumerate(body):
        if ln.startswith("s_barrier"):
            back = body[max(0, i - 4):i]
            assert any("s_waitcnt" in x for x in back), (i, back)
        if DMA.match(ln):
            assert any(re.match(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi)$", x) for x in body[max(0, i - 3):i]), (ln, body[max(0, i - 3):i])
    for key in ("vgpr_spill_count", "sgpr_spill_count"):
        pass   # (spills outside the key-block loop are allowed: they go to scratch, never to the accumulator file)


def test_attn256_body_is_what_the_generator_emits(tmp_path):
    """csrc/llama_attn256_body.inc is generated (tools/gen_attn256.py) and committed: the committed file must be the generator's
    current default output, so that a change of the generator (or a hand edit of the .inc) cannot drift unnoticed."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "body.inc")
    env = {k: v for k, v in os.environ.items() if not k.startswith("A2_")}
    env["A2_OUT"] = out
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_attn256.py")], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    assert open(out).read() == open(os.path.join(root, "llamarec_amd", "csrc", "llama_attn256_body.inc")).read()
