#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc's device assembly, and a dependency-free REPLAY of that mix as a probe kernel.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 --cuda-device-only -S \
          baler_amd/csrc/bf16_train.hip -o /tmp/bf16_train.s
    python tools/isa_mix.py /tmp/bf16_train.s 'bf16_train_kernelILi24ELi15ELi0E' [--emit tools/probe/mix_replay_part0.hip]

The loop = every basic block the assembly marks `in Loop: Header=<the header with the most instructions>`.  The replay keeps the ORDER of
instruction classes and replaces every operand by registers that nothing depends on (MFMAs rotate over 16 accumulator quads, VALU
results rotate over 24 scratch registers, LDS / buffer loads rotate over 8 quads and wait only through counted s_waitcnt that leave
8 in flight): what the hardware can issue for THIS multiset when no instruction waits for another one's result."""
import collections
import re
import sys


def kernel_lines(path, needle):
    out, on = [], False
    for line in open(path):
        if not on:
            if re.match(r"^_Z\S+:", line) and needle in line:
                on = True
            continue
        if line.startswith(".Lfunc_end"):
            break
        out.append(line.rstrip("\n"))
    return out


def loop_body(lines):
    """-> instructions (mnemonic + operands) of the biggest loop, in program order."""
    hdr_count = collections.Counter()
    cur = None
    blocks = []          # [label, header or None, [instructions]]
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cmt = m.group(2) or ""
            h = re.search(r"Header=BB(\d+_\d+)", cmt)
            own = "Loop Header" in cmt
            cur = [m.group(1), ("BB" + h.group(1)) if h else (m.group(1)[2:] if own else None), []]
            blocks.append(cur)
            continue
        if cur is None:
            cur = ["entry", None, []]
            blocks.append(cur)
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur[2].append(s.split(";")[0].strip())
    for lab, h, ins in blocks:
        if h:
            hdr_count[h] += len(ins)
    if not hdr_count:
        return []
    top = hdr_count.most_common(1)[0][0]
    seq = []
    for lab, h, ins in blocks:
        if h == top:
            seq += ins
    return seq


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt_pk_bf16"): return "valu_cvt_pk"
    if op.startswith("v_cmp"): return "valu_cmp"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read_b128") or op.startswith("ds_read2_b64"): return "ds_read_b128"
    if op.startswith("ds_read_b64_tr"): return "ds_read_tr"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read_other"
    if op.startswith("ds_write_b128") or op.startswith("ds_write2_b64"): return "ds_write_b128"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write_other"
    if op.startswith("ds_"): return "ds_other"
    if op.startswith("buffer_load") or op.startswith("global_load"): return "vmem_load"
    if op.startswith("buffer_store") or op.startswith("global_store"): return "vmem_store"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


THREE_SRC = ("v_fma", "v_med3", "v_max3", "v_min3", "v_bfi", "v_perm", "v_lshl_add", "v_add3", "v_mad", "v_and_or", "v_lshl_or", "v_or3",
             "v_maximum3", "v_minimum3", "v_cndmask")


def emit(seq, path, name, drop=(), waves=4):
    """The replay: one asm volatile block, the class sequence of `seq` with independent operands (classes in `drop` left out)."""
    out = []
    k = collections.Counter()
    nq = 16 if waves <= 4 else 12          # accumulator quads the MFMAs rotate over; v_accvgpr reads come from the registers behind them
    na = 96 if waves <= 4 else 64          # (8 waves per workgroup = 2 per SIMD: 256 registers per wave, v60..v163 + a0..a63)
    for ins in seq:
        c = classify(ins)
        if c in drop or (c.startswith("valu") and "valu_all" in drop) or (c.startswith("ds_") and "lds" in drop) or \
                (c.startswith("vmem") and "vmem" in drop):
            continue
        i = k[c]
        k[c] += 1
        if c == "mfma":
            # the kernel's own MFMA shape on independent accumulators (16 quads / 8 octets in turn)
            op = ins.split()[0]
            if op.startswith("v_mfma_f64"):
                a = 8 * (i % (nq // 2))
                out.append(f"{op} a[{a}:{a + 7}], v[64:65], v[68:69], a[{a}:{a + 7}]")
            elif op.endswith("_f32") and not op.endswith("bf16"):      # v_mfma_f32_16x16x4_f32, v_mfma_f32_4x4x1_16b_f32: one register per operand
                a = 4 * (i % nq)
                out.append(f"{op} a[{a}:{a + 3}], v64, v68, a[{a}:{a + 3}]")
            else:
                a = 4 * (i % nq)
                out.append(f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[64:67], v[68:71], a[{a}:{a + 3}]")
        elif c == "acc_mov":
            out.append(f"v_accvgpr_read_b32 v{72 + i % 24}, a{4 * nq + i % (na - 4 * nq)}")
        elif c == "valu_pk":
            d = 72 + 2 * (i % 12)
            out.append(f"v_pk_mul_f32 v[{d}:{d + 1}], v[96:97], v[98:99]")
        elif c == "valu_cvt_pk":
            out.append(f"v_cvt_pk_bf16_f32 v{72 + i % 24}, v96, v97")
        elif c == "valu_cmp":
            out.append("v_cmp_lt_f32 vcc, v96, v97")
        elif c == "valu":
            three = ins.split()[0].startswith(THREE_SRC)
            out.append(f"v_fma_f32 v{72 + i % 24}, v96, v97, v98" if three else f"v_mul_f32 v{72 + i % 24}, v96, v97")
        elif c in ("ds_read_b128", "ds_read_tr", "ds_read_other"):
            d = 100 + 4 * (i % 8)
            if c == "ds_read_b128":
                out.append(f"ds_read_b128 v[{d}:{d + 3}], v60")
            elif c == "ds_read_tr":
                out.append(f"ds_read_b64_tr_b16 v[{d}:{d + 1}], v60")
            else:
                out.append(f"ds_read_b64 v[{d}:{d + 1}], v60")
        elif c in ("ds_write_b128", "ds_write_other", "ds_other"):
            out.append("ds_write_b128 v61, v[96:99]" if c == "ds_write_b128" else "ds_write_b64 v61, v[96:97]")
        elif c == "vmem_load":
            d = 132 + 4 * (i % 8)
            out.append(f"buffer_load_dwordx4 v[{d}:{d + 3}], v62, s[8:11], 0 offen")
        elif c == "vmem_store":
            out.append("buffer_store_dwordx4 v[96:99], v63, s[8:11], 0 offen")
        elif c == "barrier":
            out.append("s_barrier")
        elif c == "waitcnt":
            # a counted wait that leaves 8 of each kind in flight: the queues stay bounded, nothing waits for a fresh result
            out.append("s_waitcnt vmcnt(8) lgkmcnt(8)")
        elif c == "salu":
            out.append("s_add_u32 s20, s20, 1")
        # branches / other: dropped
    body = " \\n\\t\"\n        \"".join(out)
    clob = ", ".join([f'"v{i}"' for i in range(60, 164)] + [f'"a{i}"' for i in range(0, na)])
    src = f'''// GENERATED by tools/isa_mix.py from the main loop of {name}: the same instruction-class sequence, every dependency removed.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mix_replay THIS_FILE && /tmp/mix_replay
// One workgroup of {waves} waves per CU (as the kernel runs); LDS reads / writes at lane * 16, buffer loads from an L2-resident MB.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__({64 * waves}) replay(const float *buf, int iters, unsigned long long *out) {{
    extern __shared__ float lds[];
    const unsigned long long bp = (unsigned long long)buf;
    const unsigned int r4[4] = {{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bp),
                                (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(bp >> 32) & 0xffffu)), 1u << 20, 0x00020000u}};
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x & 255] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {{
        asm volatile(
        "s_mov_b32 s8, %0 \\n\\t"
        "s_mov_b32 s9, %1 \\n\\t"
        "s_mov_b32 s10, %2 \\n\\t"
        "s_mov_b32 s11, %3 \\n\\t"
        "v_mov_b32 v60, %4 \\n\\t"
        "v_mov_b32 v61, %5 \\n\\t"
        "v_mov_b32 v62, %4 \\n\\t"
        "v_mov_b32 v63, %5 \\n\\t"
        "{body} \\n\\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0) \\n\\t"
        :: "s"(r4[0]), "s"(r4[1]), "s"(r4[2]), "s"(r4[3]), "v"(lane * 16), "v"(16384 + lane * 16)
        : "memory", "vcc", "s8", "s9", "s10", "s11", "s20", {clob});
    }}
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}}
int main() {{
    float *buf; unsigned long long *out;
    hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
    hipMalloc(&out, 256 * 8);
    const int iters = 200;
    hipFuncSetAttribute((const void *)replay, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) {{
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(replay, dim3(256), dim3({64 * waves}), 65536, 0, buf, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        printf("{name}: %.2f us per iteration (wall, 256 workgroups x {waves} waves), %.0f shader cycles per iteration (mean over workgroups)\\n",
               1e3 * ms / iters, s / 256 / iters);
    }}
    return 0;
}}
'''
    open(path, "w").write(src)


if __name__ == "__main__":
    path, needle = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, needle)
    seq = loop_body(lines)
    cnt = collections.Counter(classify(i) for i in seq)
    print(f"{needle}: {len(lines)} lines, main loop {len(seq)} instructions")
    for k_, v_ in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print(f"  {k_:16s} {v_:6d}")
    nm = cnt["mfma"]
    valu = cnt["valu"] + cnt["valu_pk"] + cnt["valu_cvt_pk"] + cnt["valu_cmp"] + cnt["acc_mov"]
    if nm:
        print(f"  VALU per MFMA {valu / nm:.2f}; LDS reads per MFMA {(cnt['ds_read_b128'] + cnt['ds_read_tr'] + cnt['ds_read_other']) / nm:.2f}; "
              f"LDS writes per MFMA {(cnt['ds_write_b128'] + cnt['ds_write_other']) / nm:.2f}; barriers {cnt['barrier']}")
    if "--emit" in sys.argv:
        drop = tuple(sys.argv[sys.argv.index("--drop") + 1].split(",")) if "--drop" in sys.argv else ()
        waves = int(sys.argv[sys.argv.index("--waves") + 1]) if "--waves" in sys.argv else 4
        emit(seq, sys.argv[sys.argv.index("--emit") + 1], needle + (" without " + "+".join(drop) if drop else ""), drop, waves)
        print("wrote", sys.argv[sys.argv.index("--emit") + 1])
