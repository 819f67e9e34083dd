"""Experiment builds of libqf_hip.so (round 4): textual variants of ONE csrc file each, compiled beside the package.

    python tools/experiments/variants.py NAME [NAME ...]      # builds tools/experiments/_build/libqf_NAME.so
    python tools/experiments/variants.py --list

A variant = (source file, [(anchor text, replacement text[, occurrences, which]), ...], extra hipcc flags).  The anchor
must occur exactly once (or `occurrences` times) in the product source (asserted: a variant that no longer applies fails loudly instead of silently measuring the
product kernel); the patched copy is compiled, the other objects are the product build's (csrc/_obj), misc.cpp is
rebuilt with the experiment ABI offset (+1000), so the library only loads with QF_HIP_LIBRARY_EXPERIMENT=1 and can never
be mistaken for -- or left behind as -- the product library.  hipcc cross-compiles gfx950 in the build container; the
.so travels to the GPU box with gpurun (git-ignored).  The PRODUCT sources are never touched.

Experiment kernels may return garbage VALUES; only their timing and counters mean anything.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "quadraturefields_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
EXTRA = {"exact.hip": ["-ffp-contract=off"]}

# ----------------------------------------------------------------------------------------------------------------------
# field_kernel: where does its 1.32 ms go?  
_NO_MLP = ("        // ---- base MLP 32 -> 64 (ReLU) -> 16\n", """        // EXPERIMENT no_mlp: the gathers and the blend stay live (every lane's eight features reach an output through two
        // cross-quartet adds), the whole MFMA chain and the heads are gone
        {
            float s_ = ((feat[0] + feat[1]) + (feat[2] + feat[3])) + ((feat[4] + feat[5]) + (feat[6] + feat[7]));
            s_ += __shfl_xor(s_, 16, 64);
            s_ += __shfl_xor(s_, 32, 64);
            if (HEAD == QF_HEAD_NGP || HEAD == QF_HEAD_SG) s_ += a.dirs[pt * 3 + 0] + a.dirs[pt * 3 + 1] + a.dirs[pt * 3 + 2];
            if (g == 0 && valid) {
                if (a.sigma) a.sigma[pt] = selector ? s_ : 0.0f;
                if (a.rgb) { a.rgb[pt * 3 + 0] = s_; a.rgb[pt * 3 + 1] = s_; a.rgb[pt * 3 + 2] = s_; }
            }
            continue;
        }
        // ---- base MLP 32 -> 64 (ReLU) -> 16
""")
_NO_GATHER = ("            for (int c = 0; c < 8; ++c) val[j][c] = a.table[idx[c]];\n",
              "            // EXPERIMENT no_gather: the index arithmetic stays live, the table is never read\n"
              "            for (int c = 0; c < 8; ++c) val[j][c] = make_float2((float)(idx[c] & 1023u) * 1e-3f, frac[j][c % 3]);\n",
              2, 0)            # the line occurs in field_kernel (first) and in deform_kernel: field_kernel's

# ----------------------------------------------------------------------------------------------------------------------
# bvh8_traverse_kernel: what is a ray's traversal made of?  
_TRAV_STATS_DECL = ("struct TravArgs {\n", """// EXPERIMENT trav_stats: [0] rays, [1] node steps (per octet), [2] leaf steps (per octet), [3] triangles tested,
// [4] node steps (per WAVE: iterations of the node loop with at least one octet in it), [5] leaf steps per wave,
// [6] pages beyond the first, [7] hits accepted into a list
__device__ unsigned long long qf_trav_stats[8];
extern "C" int qf_trav_stats_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(qf_trav_stats), sizeof(qf_trav_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(qf_trav_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define QF_STAT_OCT(k, v) do { if (j == 0) atomicAdd(&qf_trav_stats[k], (unsigned long long)(v)); } while (0)
#define QF_STAT_WAVE(k) do { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&qf_trav_stats[k], 1ull); } while (0)
struct TravArgs {
""")
_TRAV_STATS_RAY = ("    for (int page = 0; page < kMaxPages; ++page) {\n        list.count = 0;\n",
                   "    QF_STAT_OCT(0, 1);\n    for (int page = 0; page < kMaxPages; ++page) {\n        if (page) QF_STAT_OCT(6, 1);\n        list.count = 0;\n")
_TRAV_STATS_NODE = ("                const float4 *np = nodes + (size_t)cur * 16 + j * 2;\n",
                    "                QF_STAT_OCT(1, 1); QF_STAT_WAVE(4);\n                const float4 *np = nodes + (size_t)cur * 16 + j * 2;\n")
_TRAV_STATS_LEAF = ("                const int first = packed >> 3, cnt = (packed & 7) + 1;\n                bool h = false;\n",
                    "                const int first = packed >> 3, cnt = (packed & 7) + 1;\n                QF_STAT_OCT(2, 1); QF_STAT_OCT(3, cnt); QF_STAT_WAVE(5);\n                bool h = false;\n")
_TRAV_STATS_HIT = ("                if (m8) {\n                    const int n = __popc(m8);\n                    if (list.count + n <= Kc) {\n",
                   "                if (m8) {\n                    const int n = __popc(m8);\n                    QF_STAT_OCT(7, n);\n                    if (list.count + n <= Kc) {\n")

# ----------------------------------------------------------------------------------------------------------------------
# texture_shade_packed_kernel: record stride / order and the LDS code tables
_TEX_DECL = ("constexpr int kTexelRecord = QF_TEXEL_RECORD_BYTES;\n", """constexpr int kTexelRecord = QF_TEXEL_RECORD_BYTES;
// EXPERIMENT: where texel px = y * size + x lives in the record array
__device__ __forceinline__ int64_t exp_rec_off(int64_t px, int size)
{
#if defined(QF_EXP_TEX_TILED) || defined(QF_EXP_TEX_MORTON)
    const int y = (int)(px / size), x = (int)(px - (int64_t)y * size);
#if defined(QF_EXP_TEX_MORTON)
    const int xi = x & 7, yi = y & 7;
    const int in = (xi & 1) | ((yi & 1) << 1) | ((xi & 2) << 1) | ((yi & 2) << 2) | ((xi & 4) << 2) | ((yi & 4) << 3);
#else
    const int in = (y & 7) * 8 + (x & 7);
#endif
    px = ((int64_t)(y >> 3) * (size >> 3) + (x >> 3)) * 64 + in;
#endif
    return px * QF_EXP_TEX_STRIDE;
}
""")
_TEX_PACK_DST = ("        uint4 *dst = reinterpret_cast<uint4 *>(records + px * kTexelRecord);\n",
                 "        uint4 *dst = reinterpret_cast<uint4 *>(records + exp_rec_off(px, t.size));\n")
_TEX_PACK_N = ("        for (int k = 0; k < kTexelRecord / 16; ++k) dst[k] = rec.q[k];\n",
               "        for (int k = 0; k < QF_EXP_TEX_STRIDE / 16; ++k) dst[k] = rec.q[k];\n")
_TEX_SHADE_SRC = ("        const uint4 *src = reinterpret_cast<const uint4 *>(records + px * kTexelRecord);\n",
                  "        const uint4 *src = reinterpret_cast<const uint4 *>(records + exp_rec_off(px, size));\n")
_TEX = [_TEX_DECL, _TEX_PACK_DST, _TEX_PACK_N, _TEX_SHADE_SRC]

# (cos, sin) of the azimuth code and (sin, cos) of the elevation code as ONE 8-byte LDS read each (ds_read_b64: 64 banks)
_TEX_LDS_DECL = ("    __shared__ float s_sigma[256], s_col[256], s_caz[256], s_saz[256], s_sel[256], s_cel[256], s_lam[256];\n",
                 "    __shared__ float s_sigma[256], s_col[256], s_lam[256];\n    __shared__ float2 s_az[256], s_el[256];\n")
_TEX_LDS_FILL = ("        s_caz[c] = cosf(az);\n        s_saz[c] = sinf(az);\n        s_sel[c] = sinf(el);\n        s_cel[c] = cosf(el);\n",
                 "        s_az[c] = make_float2(cosf(az), sinf(az));\n        s_el[c] = make_float2(sinf(el), cosf(el));\n")
_TEX_LDS_USE = ("                const float se = s_sel[c_el];\n                const float x0 = s_caz[c_az] * se, x1 = s_saz[c_az] * se, x2 = s_cel[c_el];\n",
                "                const float2 az2 = s_az[c_az], el2 = s_el[c_el];\n                const float se = el2.x;\n"
                "                const float x0 = az2.x * se, x1 = az2.y * se, x2 = el2.y;\n")
_TEX_LDS = [_TEX_LDS_DECL, _TEX_LDS_FILL, _TEX_LDS_USE]

# ----------------------------------------------------------------------------------------------------------------------
# field_kernel_bf16 (configs[2]): the same two stubs, and the x-neighbour pair of a cell as ONE 8-byte request when its two
# rows are adjacent
_B_NO_MLP = ("        // ---- base MLP\n", """        // EXPERIMENT bf16_no_mlp
        {
            float s_ = ((feat[0] + feat[1]) + (feat[2] + feat[3])) + ((feat[4] + feat[5]) + (feat[6] + feat[7]));
            s_ += __shfl_xor(s_, 16, 64);
            s_ += __shfl_xor(s_, 32, 64);
            if (HEAD == QF_HEAD_NGP || HEAD == QF_HEAD_SG) s_ += a.dirs[pt * 3 + 0] + a.dirs[pt * 3 + 1] + a.dirs[pt * 3 + 2];
            if (g == 0 && valid) {
                if (a.sigma) a.sigma[pt] = selector ? s_ : 0.0f;
                if (a.rgb) { a.rgb[pt * 3 + 0] = s_; a.rgb[pt * 3 + 1] = s_; a.rgb[pt * 3 + 2] = s_; }
            }
            continue;
        }
        // ---- base MLP
""")
_B_GATHER = "            for (int c = 0; c < 8; ++c) raw[j][c] = a.table[idx[c]];\n"
_B_NO_GATHER = (_B_GATHER, "            for (int c = 0; c < 8; ++c) raw[j][c] = (idx[c] & 1023u) << 20;      // EXPERIMENT bf16_no_gather\n")
_B_XPAIR = (_B_GATHER, """            // EXPERIMENT bf16_xpair: corners c, c+1 are the cell's x-neighbours; adjacent rows (always on a dense level,
            // for even cx on a hashed one) come with one unaligned 8-byte load
            for (int c = 0; c < 8; c += 2) {
                typedef uint32_t u32x2u_ __attribute__((ext_vector_type(2), aligned(4)));
                const uint32_t i0 = idx[c], i1 = idx[c + 1];
                const uint32_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
                const bool adj = hi - lo == 1u;
                u32x2u_ v = {0u, 0u};
                uint32_t s0 = 0u, s1 = 0u;
                if (adj) v = *reinterpret_cast<const u32x2u_ *>(a.table + lo);
                else { s0 = a.table[i0]; s1 = a.table[i1]; }
                raw[j][c] = adj ? (i0 < i1 ? v.x : v.y) : s0;
                raw[j][c + 1] = adj ? (i0 < i1 ? v.y : v.x) : s1;
            }
""")

# (trav_unordered -- any hit child next instead of the nearest -- was measured here in round 4: frame 0.961 -> 0.917 ms,
# 2^17 random rays 0.549 -> 0.528; it is in the product now: bvh8_traverse_kernel<kOrdered>, chosen per mesh)

# the nearest-first traversal on every mesh (round 3's behaviour), to compare instruction counts with the product's choice
_TRAV_ORDERED = ("    const bool ordered = bvh->depth_complexity >= 0.5f * (float)max_hits;\n",
                 "    const bool ordered = true;       // EXPERIMENT trav_ordered\n")

VARIANTS = {
    "trav_ordered": ("exact.hip", [_TRAV_ORDERED], []),
    "bf16_no_mlp": ("field_eval_bf16.hip", [_B_NO_MLP], []),
    "bf16_no_gather": ("field_eval_bf16.hip", [_B_NO_GATHER], []),
    "bf16_xpair": ("field_eval_bf16.hip", [_B_XPAIR], []),
    "tex_stride48": ("exact.hip", _TEX, ["-DQF_EXP_TEX_STRIDE=48"]),
    "tex_tiled64": ("exact.hip", _TEX, ["-DQF_EXP_TEX_STRIDE=64", "-DQF_EXP_TEX_TILED"]),
    "tex_tiled48": ("exact.hip", _TEX, ["-DQF_EXP_TEX_STRIDE=48", "-DQF_EXP_TEX_TILED"]),
    "tex_morton48": ("exact.hip", _TEX, ["-DQF_EXP_TEX_STRIDE=48", "-DQF_EXP_TEX_MORTON"]),
    "tex_ldspair": ("exact.hip", _TEX_LDS, []),
    "tex_ldspair_tiled48": ("exact.hip", _TEX + _TEX_LDS, ["-DQF_EXP_TEX_STRIDE=48", "-DQF_EXP_TEX_TILED"]),
    "no_mlp": ("field_eval.hip", [_NO_MLP], []),
    "no_gather": ("field_eval.hip", [_NO_GATHER], []),
    "trav_stats": ("exact.hip", [_TRAV_STATS_DECL, _TRAV_STATS_RAY, _TRAV_STATS_NODE, _TRAV_STATS_LEAF, _TRAV_STATS_HIT], []),
}


def _run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise SystemExit("failed: %s\n%s" % (" ".join(cmd), p.stderr))
    if p.stderr.strip():
        sys.stderr.write(p.stderr)


def build(name):
    src, edits, flags = VARIANTS[name]
    sys.path.insert(0, ROOT)
    from quadraturefields_amd import build as product
    product.build()                                     # the product objects (csrc/_obj), up to date
    os.makedirs(OUT, exist_ok=True)
    text = open(os.path.join(CSRC, src)).read()
    for edit in edits:
        anchor, repl = edit[0], edit[1]
        expect, which = (edit[2], edit[3]) if len(edit) == 4 else (1, 0)       # (occurrences expected, the one to replace)
        if text.count(anchor) != expect:
            raise SystemExit(f"variant {name}: anchor occurs {text.count(anchor)} times in {src} (expected {expect}):\n{anchor}")
        at = -1
        for _ in range(which + 1):
            at = text.index(anchor, at + 1)
        text = text[:at] + repl + text[at + len(anchor):]
    patched = os.path.join(OUT, f"{name}_{src}")
    open(patched, "w").write(text)
    stem = os.path.splitext(src)[0]
    obj = os.path.join(OUT, f"{name}_{stem}.o")
    hipcc = "/opt/rocm/bin/hipcc"
    _run([hipcc] + FLAGS + EXTRA.get(src, []) + flags + ["-x", "hip", "-c", patched, "-o", obj])
    misc = os.path.join(OUT, "misc_exp.o")
    _run([hipcc] + FLAGS + ["-DQF_ABI_VERSION_OFFSET=1000", "-x", "hip", "-c", os.path.join(CSRC, "misc.cpp"), "-o", misc])
    others = [os.path.join(product.OBJ_DIR, f) for f in sorted(os.listdir(product.OBJ_DIR))
              if f.endswith(".o") and f not in (stem + ".o", "misc.o")]
    lib = os.path.join(OUT, f"libqf_{name}.so")
    _run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, obj, misc] + others)
    print("built", lib)
    return lib


if __name__ == "__main__":
    if "--list" in sys.argv or len(sys.argv) < 2:
        print("\n".join(VARIANTS))
    else:
        for n in sys.argv[1:]:
            build(n)
