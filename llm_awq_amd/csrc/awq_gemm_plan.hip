// Prefill GEMM on cdna4-interleaved weights: the TILE PLAN and the launch dispatch (bf16 and fp16, gfx950).
// Replaces the tile table + dispatch of gemm_forward_cuda_new (reference awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:1126-1236)
// for the layout the rewritten repacker emits.  The kernels live in awq_gemm_v6.hip (256 x 256 / 256 x 192 tiles of prompts with >= 256
// rows, W4 and w3c) and awq_gemm_v4n.hip (256 x 128 tiles, masked single row tile, split-K; knob gemm_v6 = 0).  (Round 1's own K loop of this file -- a compiler-scheduled 256 x (128 NSL) tile -- was retired in round 3: v4 / v4n
// issue the same products in the same order and were bit-identical to it in every test.)
#include <string.h>

#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

namespace {
constexpr int TM = 256;
constexpr double kNarrowRate = 0.80;  // 256 x 128 tiles (awq_gemm_v4n.hip) vs 256 x 256 (awq_gemm_v6.hip) at equal chip fill (0.83 against awq_gemm_v4.hip, profiles/r01_gemm_v4.txt)
int g_small_m = 1;  // knob gemm_small_m: 0 = the prefill GEMM only takes m >= 256 (see gemm_cdna4_v3_takes)
int g_splitk = 1;  // narrow tiles: split K over blocks when the tiles fill less than half of the chip and a workspace is given
int g_v6 = 1;  // 1 (default): 256-wide tiles of m >= 256 run awq_gemm_v6.hip (one software-pipelined wave per SIMD); 0 (tests): awq_gemm_v4n.hip's 128-wide tiles
int g_tile_n = 0;  // knob gemm_tile_n: 128 / 256 force one tile width for callers that pass tile_n = 0 (tests of a specific kernel)
int g_v6_192 = 1;  // knob gemm_v6_192: 0 = no 192-wide blocks in the tile plan
int g_v6_pair = 1;  // knob gemm_v6_pair: K split over pairs of 256 x 256 blocks where the 256-wide tiles fill at most half the chip (needs the workspace)
int g_v6_szh = 1;  // knob gemm_v6_szh: 1 = v6 dequantises in the f16-mantissa form when the caller hands its sz_half buffer (QuantLlamaMLP's gate/up launch): -40 VALU per K tile; neutral in round 2, +0.1-0.3 % at M = 2048 / +0.5-1.2 % at M = 4096 on the power-limited round-4 loop (profiles/r04_v6_szh_ab.txt)
int g_v4 = 1;  // knob gemm_v4: 0 = the tile kernels take no m below 256 (the skinny kernel serves 9 .. 255 rows); the loop it once selected is gone
void launch_wide(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
                 int n_end, int dtype, hipStream_t st, int bits, int epi, const void* szh = nullptr, int tile_n = 256) {
  if (g_v6 && m >= 256) {  // (szh: the caller's sz_half side buffer, reported exact for this layer -> the f16-mantissa dequant form, every block width)
    if (szh != nullptr && bits == 4 && g_v6_szh) launch_gemm_cdna4_v6(x, qw, szh, bias, out, m, n, k, n_begin, n_end, dtype, st, bits, epi, 1, tile_n);
    else launch_gemm_cdna4_v6(x, qw, szp, bias, out, m, n, k, n_begin, n_end, dtype, st, bits, epi, 0, tile_n);
    return;
  }
  // knob gemm_v6 = 0 (tests: the second implementation v6 is held against): the same columns on awq_gemm_v4n.hip's 256 x 128 tiles -- the same
  // products in the same K order.  (Round 2's 256 x 256 tile of that loop, awq_gemm_v4.hip, was removed in round 4.)
  launch_gemm_cdna4_v4n(x, qw, szp, bias, out, m, n, k, n_begin, n_end, dtype, nullptr, 0, st, bits, epi);
}
// knob gemm_v6_128 (default 1, round 3): the 128-wide tiles of m >= 256 that awq_gemm_v4n.hip would run UNSPLIT go to awq_gemm_v6.hip with
// two slabs per wave (256 x 128 blocks on the one-wave-per-SIMD loop: +4.3 ... +6.8 % on the M = 2048 prefill pass, o_proj / down_proj /
// the gate-up remainder, profiles/r03_v6_128.txt; bit-identical products and K order).  Split-K launches (short prompts that under-fill
// the chip) and m < 256 (masked single row tile) stay on awq_gemm_v4n.hip, W3 tiles too.
int g_v6_128 = 1;
// 1 = awq_gemm_v6.hip (NS = 2), 0 = awq_gemm_v4n.hip unsplit, ks >= 2 = awq_gemm_v4n.hip split into ks K ranges
int narrow_kernel(int m, int n_cols, int k, int bits, bool has_ws, int epi) {
  const size_t wsb = g_splitk && has_ws && epi == 0 ? gemm_v4n_workspace_bytes(m, n_cols, k) : 0;
  if (wsb > 0) return (int)(wsb / ((size_t)((m + TM - 1) / TM) * ((n_cols + 127) / 128) * TM * 128 * 4));
  return g_v6_128 && g_v6 && bits == 4 && m >= TM ? 1 : 0;
}
void launch_narrow(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
                   int n_end, int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits, int epi, const void* szh = nullptr) {
  const bool use_szh = szh != nullptr && bits == 4 && g_v6_szh;
  // columns whose 256-wide tiles fill at most half the chip: pairs of 256 x 256 blocks, each half of K and half of the rows (needs the workspace)
  if (g_v6 && g_v6_pair && g_splitk && !g_tile_n && ws != nullptr && m >= TM &&
      launch_gemm_cdna4_v6_pair(x, qw, use_szh ? szh : szp, bias, out, m, n, k, n_begin, n_end, dtype, ws, ws_bytes, st, bits, epi, use_szh ? 1 : 0) == 0)
    return;
  if (narrow_kernel(m, n_end - n_begin, k, bits, ws != nullptr, epi) == 1) {
    launch_gemm_cdna4_v6(x, qw, use_szh ? szh : szp, bias, out, m, n, k, n_begin, n_end, dtype, st, bits, epi, use_szh ? 1 : 0, 128);
    return;
  }
  launch_gemm_cdna4_v4n(x, qw, szp, bias, out, m, n, k, n_begin, n_end, dtype, g_splitk ? ws : nullptr, ws_bytes, st, bits, epi);
}
}  // namespace

namespace {
int g_moe_v4 = 1;  // knob moe_v4: 0 = every grouped batch above 8 rows on the 128 x 128 grouped kernel (tests: the second implementation of the grouped skinny / v6 kernels)
}
bool moe_v4_enabled() { return g_moe_v4 != 0; }
namespace {
int g_moe_v6 = 1;  // grouped prefill GEMM (>= 256 sorted rows) on the v6 tile (awq_gemm_v6.hip); 0: the 128 x 128 grouped kernel
}
bool moe_v6_enabled() { return g_moe_v6 != 0; }

int gemm_v3_tune_set(const char* key, int value) {
  if (!strcmp(key, "moe_v6")) g_moe_v6 = value;
  else if (!strcmp(key, "moe_tail")) moe_v6_set_tail(value);
  else if (!strcmp(key, "gemm_v4")) g_v4 = value;
  else if (!strcmp(key, "gemm_v6")) {  // units: 0 off, 1 = the 256-wide tiles, 2 = every tile; tens (probe builds): timing-only probe of the kernel
    g_v6 = value % 10;
    gemm_v6_set_probe(value / 10);
  }
  else if (!strcmp(key, "gemm_tile_n")) g_tile_n = value;
  else if (!strcmp(key, "gemm_v6_szh")) g_v6_szh = value;
  else if (!strcmp(key, "gemm_v6_pair")) g_v6_pair = value;
  else if (!strcmp(key, "gemm_v6_pair_lead")) gemm_v6_set_pair_lead(value);
  else if (!strcmp(key, "gemm_v6_pair_min_nit")) gemm_v6_set_pair_min_nit(value);
  else if (!strcmp(key, "gemm_v6_192")) g_v6_192 = value;
  else if (!strcmp(key, "gemm_v6_128")) g_v6_128 = value;
  else if (!strcmp(key, "moe_v4")) g_moe_v4 = value;
  else if (!strcmp(key, "gemm_small_m")) g_small_m = value;
  else if (!strcmp(key, "gemm_splitk_cap")) g_v4n_ksplit_cap = value;
  else if (!strcmp(key, "gemm_splitk")) {  // 0 = off, 1 = auto, n > 1 = force n K ranges
    g_splitk = value;
    g_v4n_ksplit_force = value > 1 ? value : 0;
  }
  else return -1;
  return 0;
}

// tile_n: 0 = pick by chip fill (256 CUs, one block per CU), 128 / 256 = force one width for the whole matrix.
// In auto mode a matrix whose 256-wide tile count is k full rounds plus a partial one runs the full rounds with 256-wide
// tiles and the remaining weight rows with 128-wide tiles in a second launch when that is faster.
namespace {
struct Plan {
  int mode;        // 0 = all 256-wide, 1 = all 128-wide, 2 = 256-wide for [0, cols_main * 256) + 128-wide for the rest, 3 = all 192-wide
  long cols_main;
};
constexpr double k192Rate = 0.97;  // 256 x 192 blocks of awq_gemm_v6.hip (three slabs per wave) vs its 256 x 256 blocks at equal chip fill (round-6 kernel stats: a round of them is 91 us against 0.75 x 108 = 81, i.e. 0.89 -- with that value only qkv at 4096 rows changes plan, to one 256-wide round + 256 two-slab blocks, and the M = 4096 pass moves by 0.1-0.3 %: tools/EXPERIMENTS.md)
Plan plan_tiles(int m, int n, int tile_n, bool allow192 = false) {
  if (tile_n == 128) return {1, 0};
  if (tile_n == 256) return {0, 0};
  if (tile_n == 192) return {allow192 ? 3 : 0, 0};
  const long tiles_m = (m + TM - 1) / TM;
  auto rounds = [](long t) { return (double)((t + 255) / 256); };
  const long cols256 = (n + 255) / 256, t256 = tiles_m * cols256;
  const double cost_wide = rounds(t256);
  const double cost_narrow = rounds(tiles_m * ((n + 127) / 128)) * 0.5 / kNarrowRate;
  // mixed: as many whole rounds of 256-wide tiles as fit, the remaining rows 128-wide
  const long cols_main = (t256 / 256) * 256 / tiles_m;  // 256-wide column tiles that fill whole rounds
  double cost_mixed = 1e30;
  if (cols_main > 0 && cols_main < cols256) {
    const long n_rest = n - cols_main * 256;
    cost_mixed = rounds(tiles_m * cols_main) + rounds(tiles_m * ((n_rest + 127) / 128)) * 0.5 / kNarrowRate + 0.02;
  }
  // 192-wide blocks: three quarters of the work per block; they win where they turn a partial round into a full one
  const double cost_192 = allow192 ? rounds(tiles_m * ((n + 191) / 192)) * 0.75 / k192Rate : 1e30;
  if (cost_192 < cost_wide && cost_192 < cost_narrow && cost_192 < cost_mixed) return {3, 0};
  if (cost_mixed < cost_wide && cost_mixed < cost_narrow) return {2, cols_main};
  return {cost_narrow < cost_wide ? 1 : 0, 0};
}
}  // namespace

// Which m the prefill GEMM takes.  m >= 256 always; below that the narrow-tile kernel runs ONE row tile whose missing rows are
// not stored, at the cost of a 256-row tile whatever m is, while the skinny kernel (awq_skinny_cdna4.hip) re-streams and
// re-dequantises the weights once per 64 rows: the crossover measured on the Llama shapes (profiles/r01_small_m_sweep.txt) sits
// at m ~ 75 for K >= 8192 and ~150 for K = 4096, which m * K >= 0.6 M approximates; at m = 255 the GEMM is 1.6 - 3.1x faster.
bool gemm_cdna4_v3_takes(int m, int k) {
  if (m >= TM) return true;
  if (g_small_m == 2) return g_v4 && m > 8;  // experiments: every m the skinny kernel would take
  return g_small_m && g_v4 && m >= 72 && (long)m * k >= 600000;
}

// fp32 workspace the call below can use to split K when its tiles under-fill the chip (0 = none needed)
size_t gemm_cdna4_v3_workspace_bytes(int m, int n, int k) {
  if (!gemm_cdna4_v3_takes(m, k) || (n % 16) != 0 || (k % 128) != 0 || !g_v4 || !g_splitk) return 0;
  if (gemm_cdna4_v3_pair_plan(m, n, k)) return gemm_v6_pair_workspace_bytes(m, n, k);
  if (m < TM) return gemm_v4n_workspace_bytes(m, n, k);
  const Plan p = plan_tiles(m, n, 0);
  if (p.mode == 1) return gemm_v4n_workspace_bytes(m, n, k);
  if (p.mode == 2) {  // the columns behind the full rounds: a block-pair launch where they fill at most half the chip, else the narrow tiles' split-K scratch
    const int rest = n - (int)(p.cols_main * 256);
    return gemm_cdna4_v3_pair_plan(m, rest, k) ? gemm_v6_pair_workspace_bytes(m, rest, k) : gemm_v4n_workspace_bytes(m, rest, k);
  }
  return 0;
}

// 1 = a W4 call WITH its workspace runs as pairs of 256 x 256 blocks, each summing half of K (awq_gemm_v6.hip: gemm_cdna4_v6_pair_kernel)
int gemm_cdna4_v3_pair_plan(int m, int n, int k) { return g_v6 && g_v6_pair && g_splitk && !g_tile_n && gemm_v6_pair_takes(m, n, k) ? 1 : 0; }

size_t gemm_cdna4_v3_workspace_bytes_w3(int m, int n, int k) {
  if (m <= 8 || (n % 16) != 0 || (k % 128) != 0 || !g_splitk) return 0;
  if (gemm_cdna4_v3_pair_plan(m, n, k)) return gemm_v6_pair_workspace_bytes(m, n, k);
  if (m < TM) return gemm_v4n_workspace_bytes(m, n, k);
  const Plan p = plan_tiles(m, n, 0);
  if (p.mode == 1) return gemm_v4n_workspace_bytes(m, n, k);
  if (p.mode == 2) {
    const int rest = n - (int)(p.cols_main * 256);
    return gemm_cdna4_v3_pair_plan(m, rest, k) ? gemm_v6_pair_workspace_bytes(m, rest, k) : gemm_v4n_workspace_bytes(m, rest, k);
  }
  return 0;
}

// which tiles the prefill GEMM would launch for (m, n) of a `bits`-bit matrix: *mode = 0 all 256-wide, 1 all 128-wide, 2 = 256-wide for the
// first *cols_main column tiles + 128-wide for the rest, 3 all 192-wide; returns the number of thread blocks (0: the GEMM does not take m)
int gemm_cdna4_v3_plan(int m, int n, int bits, int* mode, int* cols_main) {
  if (m <= 8 || (n % 16) != 0) return 0;
  const bool allow192 = g_v6 != 0 && g_v6_192 != 0 && bits == 4 && m >= TM;
  const Plan p = m < TM ? Plan{1, 0} : plan_tiles(m, n, g_tile_n, allow192);
  if (mode) *mode = p.mode;
  if (cols_main) *cols_main = (int)p.cols_main;
  const long tm = (m + TM - 1) / TM;
  if (p.mode == 0) return (int)(tm * ((n + 255) / 256));
  if (p.mode == 1) return (int)(tm * ((n + 127) / 128));
  if (p.mode == 3) return (int)(tm * ((n + 191) / 192));
  return (int)(tm * p.cols_main + tm * ((n - p.cols_main * 256 + 127) / 128));
}

int gemm_cdna4_v3_narrow_kernel(int m, int n_cols, int k, int bits, int has_workspace, int epi) {
  if (m <= 8 || n_cols < 16 || (n_cols % 16) != 0 || k < 128 || (k % 128) != 0) return 0;
  return narrow_kernel(m, n_cols, k, bits, has_workspace != 0, epi);
}

int launch_gemm_cdna4_v3(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                         int tile_n, int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits, int epi, const void* szh) {
  // (w3c tiles have no skinny kernel: the masked single-row-tile path of the narrow kernel serves every m > 8)
  if (!szp || !(gemm_cdna4_v3_takes(m, k) || ((bits == 3 || epi) && m > 8)) || (n % 16) != 0 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31)) return -1;
  // tiles that fill at most half the chip and a long K: pairs of 256 x 256 blocks, each half of K, combined inside the launch (needs the workspace)
  const bool use_szh = szh != nullptr && bits == 4 && g_v6_szh;
  if (g_v6 && g_v6_pair && g_splitk && (bits == 4 || bits == 3) && (epi == 0 || epi == 2) && !tile_n && !g_tile_n && ws != nullptr &&
      launch_gemm_cdna4_v6_pair(x, qw, use_szh ? szh : szp, bias, out, m, n, k, 0, n, dtype, ws, ws_bytes, st, bits, epi, use_szh ? 1 : 0) == 0)
    return 0;
  const bool allow192 = g_v6 != 0 && g_v6_192 != 0 && bits == 4 && m >= TM;
  Plan p = m < TM ? Plan{1, 0} : plan_tiles(m, n, tile_n ? tile_n : g_tile_n, allow192);  // m < 256: only the narrow-tile kernel masks rows
  if (g_v6 >= 2) p = Plan{0, 0};  // experiments: every tile 256-wide
  if (p.mode == 3) {
    launch_wide(x, qw, szp, bias, out, m, n, k, 0, n, dtype, st, bits, epi, szh, 192);
  } else if (p.mode == 2) {
    launch_wide(x, qw, szp, bias, out, m, n, k, 0, (int)(p.cols_main * 256), dtype, st, bits, epi, szh);
    launch_narrow(x, qw, szp, bias, out, m, n, k, (int)(p.cols_main * 256), n, dtype, ws, ws_bytes, st, bits, epi, szh);
  } else if (p.mode == 1) {
    launch_narrow(x, qw, szp, bias, out, m, n, k, 0, n, dtype, ws, ws_bytes, st, bits, epi, szh);
  } else {
    launch_wide(x, qw, szp, bias, out, m, n, k, 0, n, dtype, st, bits, epi, szh);
  }
  return 0;
}

}  // namespace awq
