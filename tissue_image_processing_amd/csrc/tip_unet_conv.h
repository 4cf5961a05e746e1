// tip_unet_conv.h -- the U-Net's dense convolutions (pl.py:31-72: Conv2D 3x3 'same', Conv2DTranspose 3x3 stride 2 'same') as an
// implicit GEMM on the bf16 matrix cores with SPLIT float32 operands and float32 accumulation.
//
// Arithmetic.  gfx950's float32-input MFMA runs at the float32 VECTOR rate (157 TFLOP/s), 1/16 of the bf16 rate, and has no
// TF32-like mode.  A float32 value a splits exactly into bf16 pieces a = a0 + a1 (+ a2) + r with a0 = bf16(a), a1 = bf16(a - a0),
// a2 = bf16(a - a0 - a1): |r| <= 2^-17 |a| with two pieces, 2^-25 |a| with three.  A product of two split values is then a short
// sum of bf16 x bf16 products, each of which the matrix core forms exactly and accumulates in float32:
//     two pieces   a b ~ a0 b0 + a0 b1 + a1 b0                              (3 MFMAs; dropped terms <= 2^-15.9 |a b| in total)
//     three pieces a b ~ a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0      (6 MFMAs; dropped terms <= 2^-23.4 |a b|)
// so every term of a convolution's dot product carries a relative error below 1.6e-5 (two pieces) resp. 9e-8 (three), on top
// of the float32 accumulation that any float32 convolution has (~ sqrt(K) 6e-8 for K = 1152 ... 9216 terms: 2e-6 ... 6e-6).
// Activations travel between layers ALREADY SPLIT -- `planes` bf16 images [plane][y][x][channel]; two planes are 4 bytes per
// value, exactly a float32 -- so the main loop moves bf16 tiles and does no conversion; the producer's epilogue (bias -> ReLU ->
// BatchNorm scale / shift, all in float32 on the accumulator) does the split.  Weights are split once when the model is loaded.
//
// Decomposition.  out[pixel, n] = sum over taps t and input channels c of in[pixel + (dy_t, dx_t), c] * w[t][c][n]: a GEMM with
// M = pixels, N = output channels, K = taps * Cin.  A workgroup owns a TH x 32-pixel tile (TH = 16: eight waves, one workgroup per
// CU; TH = 8: four waves, two per CU) x 128 output channels; the K loop walks the input channels in chunks of 16 and, inside a
// chunk, the taps:
//   * the chunk's (TH + 2) x (32 + 2)-pixel halo tile of the activation is staged ONCE in LDS (zero outside the image) and every
//     tap reads its shifted window from there -- the nine taps of a 3x3 stencil re-use one staged tile;
//   * the weights of one (chunk, tap) step -- 16 x 128 values per plane -- sit in one of D + 1 LDS buffers; activations and weights
//     arrive by asynchronous buffer_load ... lds copies issued D steps / a chunk ahead (no staging registers, no vector arithmetic:
//     per-lane offsets are fixed for the kernel, the step's offset is scalar);
//   * a wave computes 64 pixels (two tile rows) x 128 channels: 2 x 4 accumulator tiles of v_mfma_f32_32x32x16_bf16, the 16
//     channels of a chunk being exactly the K of one MFMA, formed TRANSPOSED (weights = the A operand) so that a lane ends up with
//     one pixel and sixteen adjacent channels per 32-channel block -- the epilogue splits and stores without a lane exchange.
//     LDS rows are 32 bytes, unpadded, conflict-free for ds_read_b128 through an XOR swizzle applied on the copy's source side.
// Two inputs (in0 with c0 channels, then in1 with c1) are read as one concatenated tensor: the decoder's
// concatenate([upsampled, skip]) (pl.py:52) never exists in memory.  A stride-2 transposed convolution is four such
// convolutions, one per output parity class, with 4 / 2 / 2 / 1 taps and a strided output (tap lists built by the caller).
// The kernel body has the details next to the code; DESIGN.md 5.7 has the measurements (clock trace, power-limited MFMA ceiling).
#pragma once
#include "tip_internal.h"

namespace tip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float UC_F16_MAX = 65504.f;          // largest finite fp16: scaled activations saturate there (see "piece formats")

constexpr int UC_TW = 32;                     // pixel tile: TH rows (8 or 16: template parameter) x 32 columns
constexpr int UC_BN = 128;                    // output channels per workgroup
constexpr int UC_KC = 16;                     // input channels per chunk = K of one MFMA
constexpr int UC_HW = UC_TW + 2;               // halo tile: (TH + 2) rows of 34 pixels

struct ConvParams {
    const uint16_t *in0, *in1;     // split-plane activations [plane][H][W][C]
    int c0, c1;                    // channels of in0 / in1 (multiples of 16; c1 = 0 without a second input)
    int H, W;                      // input grid
    const uint16_t *w;             // packed weights [tap][chunk][nblk][plane][128][16]
    int ntaps;
    int dy[9], dx[9];              // input offset of every tap (-1, 0, 1)
    int cout;                      // multiple of 128
    const float *bias, *scale, *shift;   // scale == nullptr: bias only (Conv2DTranspose); else bias -> ReLU -> scale, shift
    uint16_t *out;                 // [plane][outH][outW][cout]
    int outH, outW, sy, sx, oy, ox;       // output pixel of input-grid pixel (y, x): (y * sy + oy, x * sx + ox)
    const float *head_w, *head_b;  // nullptr, or the network's head fused into this layer's epilogue (cout == 128, plain output mapping):
    float *head_out;               //   Conv2D(128 -> 2, 1x1) weights [2][128], bias [2] -> softmax -> float32 (2, H, W); `out` is then not written
#ifdef UC_TRACE
    unsigned long long *trace;     // diagnostic build: clock sums of block 0 / wave 0 (tools/unet_trace.py)
#endif
    int xcd_map;                   // workgroup -> (tile, channel block) mapping, see the kernel
    float acc_scale;               // fp16 pieces: the accumulator is multiplied by this (1 / (activation scale x weight scale), a power of two) before the bias
    int nmask[9];                  // fused transposed convolution (template TF): which of the four 32-channel accumulator blocks -- the four output
                                   // parity classes -- tap t feeds (bit n); `cout` then counts VIRTUAL channels [group of 32][class][32]
    uint16_t *pool_out;            // nullptr, or [plane][outH / 2][outW / 2][cout]: MaxPool2D(2) of the output (needs sy = sx = 1)
};

__device__ __forceinline__ unsigned bf16_rne_bits(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;     // round to nearest even (finite values)
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned h) { return __uint_as_float(h << 16); }
// max(v, 0) on the bit pattern (a negative float is a negative integer): one instruction, no NaN canonicalisation in front
__device__ __forceinline__ float relu_bits(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
// plain v_max_f32 (fmaxf() quiets signalling NaNs first: two more instructions per call)
__device__ __forceinline__ float fmax_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(v, v of the neighbouring lane (lane ^ 1))
__device__ __forceinline__ float fmax_pair(float v)
{
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

typedef __attribute__((address_space(3))) unsigned char lds_byte;

// Piece formats.  F16 == false: bf16 pieces (8 significand bits each; NPL = 2: three products per term, <= 2^-15.9 dropped; NPL = 3:
// six products, float32-equivalent).  F16 == true: fp16 pieces (11 significand bits each) of SCALED values -- hi = fp16(s v),
// lo = fp16(s v - hi): |s v - hi - lo| <= 2^-22 |s v| while lo is a normal fp16 number, and an ABSOLUTE 2^-25 / s below that (fp16
// subnormals are kept by v_cvt_f16_f32 and by the matrix core) -- so the same three products hi hi + hi lo + lo hi drop <= 3 x 2^-22
// of a term: float32-equivalent at the bf16x3 cost.  The price is fp16's range: s v has to stay below 65504.  Activations are
// stored with s = 2^4 (they saturate beyond |v| = 4094; the absolute floor is 2^-29), every layer's weights with their own power of
// two that puts the largest one in [2^14, 2^15); the accumulator is multiplied by acc_scale = 1 / (both) -- all powers of two, so
// the scaling itself is exact.
template <bool F16>
__device__ __forceinline__ f32x16 uc_mfma(const uint4 &a, const uint4 &b, const f32x16 &c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// rounds (a, b) to the next piece (nearest even) and keeps the remainders when more pieces follow; returns the packed word
template <bool F16>
__device__ __forceinline__ unsigned uc_piece_word(float &a, float &b, bool more)
{
    if constexpr (F16) {
        f16x2 hv;
        hv[0] = (_Float16)a;
        hv[1] = (_Float16)b;
        if (more) {
            a -= (float)hv[0];
            b -= (float)hv[1];
        }
        return __builtin_bit_cast(unsigned, hv);
    } else {
        bf16x2 hv;
        hv[0] = (__bf16)a;
        hv[1] = (__bf16)b;
        const unsigned w = __builtin_bit_cast(unsigned, hv);
        if (more) {
            const f32x2 r = f32x2{a, b} - f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
            a = r[0];
            b = r[1];
        }
        return w;
    }
}
template <bool F16>
__device__ __forceinline__ float uc_piece_value(unsigned halfword)
{
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)halfword);
    else return __uint_as_float(halfword << 16);
}
__device__ __forceinline__ float uc_sat_f16(float v) { return __builtin_amdgcn_fmed3f(v, -UC_F16_MAX, UC_F16_MAX); }

// A counted wait on the vector-memory queue followed by the workgroup barrier.  The asynchronous global -> LDS copies of
// LATER steps stay in flight across the barrier (a __syncthreads() would drain them: its fence waits for vmcnt(0) while an
// LDS-DMA is pending), so the count is the number of copy instructions this wave issued AFTER the ones it must see landed.
#ifdef UC_TRACE
__device__ unsigned long long uc_trace_wait_ticks;      // (diagnostic build only) never read: the per-wave sum lives in a register
template <int N>
__device__ __forceinline__ unsigned long long uc_wait_barrier_timed()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return t;
}
#endif
template <int N>
__device__ __forceinline__ void uc_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// NPL = bf16 pieces per value (2 or 3).
//
// LDS image (one dynamic array; 16-byte slots):  two activation buffers of A_SLOTS slots -- slot (plane * 340 + pixel) * 2 + sh --
// and D + 1 weight buffers of NPL * 256 slots -- slot (plane * 128 + n) * 2 + sh.  A row (one pixel / one output channel) is
// 16 bf16 = 32 bytes = two slots; the two halves of row j are stored SWAPPED when bit 3 of j is set (sh = half ^ ((j >> 3) & 1)):
// the 16-lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} of consecutive rows, one half each) then hit 16
// different bank quads with unpadded rows, which is what lets the tiles arrive by asynchronous ... lds copies (a wave's 64 x 16
// bytes land in 64 consecutive slots; the swizzle is applied to the per-lane SOURCE offset).
//
// TH = tile rows = 2 per wave: 8 (256 threads, two workgroups per CU) or 16 (512 threads, one per CU).  The weight tile of a step
// is shared by all the workgroup's waves, so the taller tile halves the weight bytes copied per MFMA -- and the copies are what
// the kernel is short of: with the MFMAs taken out the 8-row kernel still runs 45 % of its time (its copies pull ~9 TB/s out of
// L2), and only a third of that hides behind the matrix work.
//
// D = how many steps ahead the weight copies run (D + 1 buffers).  A copy has to cross L2 under the load of every other CU's
// copies: two steps (~1.3 us) is about its latency, so the counted wait in front of the barrier regularly stalls on it; four
// steps (the 16-row kernel has the LDS for five buffers) take it off the critical path.  D > 2 needs ntaps >= D (then the
// activation tile of the next chunk, issued at the chunk's first tap, is always older than the weights a step waits for).
//
// DA = how many chunks ahead the activation copies run (DA + 1 buffers): one is enough for a 3x3 stencil (nine steps of flight);
// with one or two taps per chunk -- the small parity classes of the transposed convolution -- the tile issued at a chunk's first
// tap is due one or two steps later and the wave sat on it, so those run two chunks ahead (DA = 2 needs D = 2).
//
// SPB = steps per barrier (2 or 3: 3x3 stencils on 16-row tiles, the step count a multiple of SPB).  Every barrier costs the matrix
// pipe a fixed few hundred clocks -- the copies' issue, the first operands' way out of LDS, the counted wait, the barrier itself (clock
// trace) -- against ~1 540 clocks of MFMA issue per step; with SPB steps between barriers the later steps' fragment reads go out behind
// the earlier steps' MFMAs and that price is paid once per SPB x 24 MFMAs of a wave.  3 SPB weight buffers: the steps of this
// iteration are read, those of the next must have landed by its end, those of the one after are issued at its start.  An iteration
// may straddle a chunk border (SPB = 2: nine taps per chunk), so the next chunk's activations are issued at the start of the first
// iteration that BEGINS inside the current chunk: every read of the previous chunk -- whose buffer they overwrite -- lies behind a
// barrier by then, and at least six steps of flight remain.
// TF = the stride-2 transposed convolution (pl.py:47) as ONE launch: the four taps are the four input offsets (0, 0), (0, -1), (-1, 0),
// (-1, -1); the four 32-channel accumulator blocks of a wave are the four output parity classes (py, px) of the SAME 32 output
// channels, and a tap feeds the classes whose kernel element it holds (nmask: 1111, 0101, 0011, 0001 -- nine products per staged
// tile instead of 4 / 2 / 2 / 1 in four launches); a workgroup covers 32 real output channels, and the epilogue scatters class n to
// output pixel (2 y + n / 2, 2 x + n % 2).  Accumulation order per output equals the four-launch form's: bit-identical results.
template <int NPL, int TH, int D, int DA = 1, int SPB = 1, bool F16 = false, bool TF = false>
__global__ void __launch_bounds__(TH * 32, (NPL == 2) ? 2 : 1) k_unet_conv(const ConvParams p)
{
#if defined(__HIP_DEVICE_COMPILE__)       // (the buffer-resource builtins exist in the device pass only; the host pass needs just the stub)
#ifdef UC_TRACE
    const unsigned long long tr_enter = __builtin_amdgcn_s_memtime();
#endif
    static_assert(DA == 1 || (DA == 2 && D == 2), "two-chunk activation prefetch: with the two-step weight schedule");
    static_assert(SPB == 1 || ((SPB == 2 || SPB == 3) && D == 4 && DA == 1 && TH == 16), "several steps per barrier: the 16-row kernel with the four-step weight schedule");
    constexpr bool PAIR = SPB > 1;
    constexpr int UC_NBBUF = PAIR ? 3 * SPB : D + 1;
    constexpr int UC_THREADS = TH * 32, UC_HP = UC_HW * (TH + 2);
    static_assert(TH == 8 || (TH == 16 && NPL == 2), "16-row tiles: two pieces (LDS)");
    static_assert(!TF || (NPL == 2 && SPB == 1), "fused transposed convolution: two pieces, one step per barrier");
    // a plane's halo tile is padded to whole waves of slots (A_PLANE): one copy instruction of one wave then serves ONE plane, and
    // the plane picks the buffer resource (scalar) instead of adding a plane stride to the 32-bit per-lane offset
    constexpr int A_GPP = (UC_HP * 2 + 63) / 64, A_PLANE = A_GPP * 64;    // wave-groups / slots per plane (20 / 1280 for 16 rows, 11 / 704 for 8)
    constexpr int A_PER = (NPL * A_PLANE + UC_THREADS - 1) / UC_THREADS;  // copy instructions per thread and chunk (5 / 6 / 9)
    constexpr int A_SLOTS = A_PER * UC_THREADS;                           // padded: every wave issues the same number of copies
    constexpr int A_BYTES = A_SLOTS * 16;
    constexpr int B_PER = NPL * 256 / UC_THREADS;                         // weight slots per thread (NPL * 256 slots)
    static_assert(B_PER * UC_THREADS == NPL * 256, "weight slots divide evenly");
    constexpr int B_BYTES = NPL * 256 * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sA = smem, *sB = smem + (DA + 1) * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: LDS bases of the copies go to M0 without a waterfall loop)
    const int tilesX = p.W / UC_TW;
    const int nblks = p.cout / UC_BN;
    // Workgroup -> (pixel tile, block of 128 output channels).  Plain order: grid (tiles, blocks) -- every workgroup in flight works
    // on the SAME channel block (its weights stay in L2) but the input is streamed from HBM once per block (up to eight times).
    // XCD order (p.xcd_map; 1-D grid of tiles x blocks, tiles % 8 == 0): the hardware deals workgroup b to XCD b % 8, so with
    // j = b / 8 the workgroups j = 0 .. nblks - 1 of one XCD are the channel blocks of ONE pixel tile: they run side by side and
    // the tile's activations come out of that XCD's L2 for all but the first of them; the weights of all blocks then stream through
    // each L2 from the memory-side cache (a layer's packed weights are at most 38 MB).
    int tile, nblk;
    if (p.xcd_map) {
        const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
        tile = (j / nblks) * 8 + xcd;
        nblk = j - (j / nblks) * nblks;
    } else {
        tile = blockIdx.x;
        nblk = blockIdx.y;
    }
    const int ty0 = (tile / tilesX) * TH, tx0 = (tile % tilesX) * UC_TW;
    const int cin = p.c0 + p.c1, nchunks = cin / UC_KC, nsteps = nchunks * p.ntaps;
    const long in_plane0 = (long)p.H * p.W * p.c0, in_plane1 = (long)p.H * p.W * p.c1;
    const int rowbase = ty0 > 0 ? ty0 - 1 : 0;                // first image row of the halo window
    const int winrows = min(p.H, ty0 + TH + 1) - rowbase;     // ... and its rows inside the image

    // ---- copy plans (fixed per thread) ------------------------------------------------------------------------------------
    // The tiles arrive by buffer_load_dwordx4 ... lds: a buffer resource (scalar registers: base, extent) + a per-lane byte offset
    // that is FIXED for the whole kernel + a scalar offset that moves with the chunk / step.  A step's copies are then a handful of
    // scalar instructions and the copy itself -- no vector arithmetic: clock counters inside the kernel (tools/unet_trace.sh)
    // showed every step opening with ~210 clocks of copy addressing in all waves, during which the matrix pipe had nothing to
    // issue, and vector instructions of one wave also delay its SIMD partner's MFMAs.  Halo pixels outside the image (and the
    // padding slots) carry an offset beyond the resource's extent: the hardware's range check returns zeros for them.
    // activation slot q = u * THREADS + tid = group g = u * (THREADS / 64) + wave (scalar), lane: plane g / A_GPP, slot inside the plane
    // -> halo pixel, stored half -> image pixel and logical half.  The buffer resource of a copy covers ONE plane's halo WINDOW (the
    // image rows rowbase .. rowbase + winrows - 1 of the input: <= (TH + 2) W C 2 bytes), so tensors of any size are addressed with
    // 32-bit offsets relative to the window.
    unsigned a_off[A_PER];
    auto plan_a = [&](int C) {
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            const int g = u * (UC_THREADS / 64) + wave, pl = g / A_GPP;
            const int rem = (g - pl * A_GPP) * 64 + lane, px = rem >> 1, half = (rem & 1) ^ ((px >> 3) & 1);
            const int hy = px / UC_HW, hx = px - hy * UC_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool inside = pl < NPL && rem < UC_HP * 2 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            a_off[u] = inside ? (unsigned)((((long)(gy - rowbase) * p.W + gx) * C + half * 8) * 2) : 0xffff0000u;     // (beyond any admissible extent -- the launcher keeps a window below 2^32 - 65536 bytes -- and the step's scalar offset cannot wrap it around)
        }
    };
    plan_a(p.c0);
    bool a_second = false;                  // the offsets are those of in1 (they depend on the channel count)
    constexpr int RSRC_FLAGS = 0x00020000;  // raw buffer, 32-bit data format (gfx9 resource word 3)
    // the window's start in every plane of in0 / in1 (scalars, no arrays: a private array of pointers would live in scratch); a copy
    // instruction's plane follows from (u, wave), and its resource words are put together at the copy -- a few scalar moves
    // the window's start in plane 0 of the CURRENT input, its plane stride and extent: scalars that switch once, when the chunks reach
    // the second input (one copy path -- two branches with their own pointers made the compiler build a scratch table of them)
    unsigned long long a_win = (unsigned long long)(p.in0 + (long)rowbase * p.W * p.c0), a_plane_bytes = (unsigned long long)in_plane0 * 2;
    int a_rec = (int)(unsigned)((long)winrows * p.W * p.c0 * 2);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)0xffffffffu, RSRC_FLAGS);
    auto copy_a = [&](int chunk, int buf) {
        const int cbase = chunk * UC_KC;
        const bool second = cbase >= p.c0;
        if (second && !a_second) {          // (once per kernel, and only when a second input exists)
            a_second = true;
            a_win = (unsigned long long)(p.in1 + (long)rowbase * p.W * p.c1);
            a_plane_bytes = (unsigned long long)in_plane1 * 2;
            a_rec = (int)(unsigned)((long)winrows * p.W * p.c1 * 2);
            if (p.c1 != p.c0) plan_a(p.c1);
        }
        const int soff = (second ? cbase - p.c0 : cbase) * 2;
        lds_byte *dst = (lds_byte *)(sA + buf * A_BYTES + wave * 64 * 16);
#pragma unroll
        for (int u = 0; u < A_PER; ++u) {
            const int pl = min((u * (UC_THREADS / 64) + wave) / A_GPP, NPL - 1);      // this wave's plane (padding groups: every lane carries the out-of-range offset)
            const unsigned long long base = a_win + (unsigned long long)pl * a_plane_bytes;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(a_rec), RSRC_FLAGS);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + u * UC_THREADS * 16, 16, a_off[u], soff, 0, 0);
        }
    };
    // weight slot q = u * THREADS + tid = (plane q / 256, n = (q % 256) / 2, stored half q & 1)
    const int bq = tid & 255;
    const unsigned b_off = (unsigned)(((tid >> 8) * (UC_BN * UC_KC) + (bq >> 1) * UC_KC + (((bq & 1) ^ ((bq >> 4) & 1))) * 8) * 2);   // byte offset inside the step's tile
    auto copy_b = [&](int chunk, int tap, int buf) {
        const int soff = ((tap * nchunks + chunk) * nblks + nblk) * (NPL * UC_BN * UC_KC * 2);        // (a layer's packed weights stay below 2 GB)
        lds_byte *dst = (lds_byte *)(sB + buf * B_BYTES + wave * 64 * 16);
#pragma unroll
        for (int u = 0; u < B_PER; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dst + u * UC_THREADS * 16, 16, b_off, soff + u * (UC_THREADS / 256) * (UC_BN * UC_KC * 2), 0, 0);
    };

    // Accumulators: D = W^T X^T -- the WEIGHT fragment is the matrix core's A operand, so a lane holds ONE pixel (column lane & 31)
    // and sixteen output channels per 32-channel block: register i of half-wave hf is row 8 (i >> 2) + 4 hf + (i & 3), and the packed
    // weight tile stores channel 16 hf + i in that row (host: _split_pack), i.e. the lane's sixteen registers are the sixteen ADJACENT
    // channels n * 32 + 16 hf + i of its pixel.
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    // prologue: activation chunk 0, weight steps 0 .. D-1
    copy_a(0, 0);
    if (DA == 2 && nchunks > 1) copy_a(1, 1);
    for (int s0 = 0; s0 < (PAIR ? 2 * SPB : D) && s0 < nsteps; ++s0) copy_b(s0 / p.ntaps, s0 % p.ntaps, s0);
    uc_wait_barrier<0>();

    const int r = lane & 31, h = lane >> 5;
    const int b_row = r * 2 + (h ^ ((r >> 3) & 1));          // slot of this lane's weight fragment inside a plane's 32 rows
    int step = 0;
    int nc = D / p.ntaps, nt = D % p.ntaps;                  // (chunk, tap) of step + D
    int buf0 = 0, buf2 = D;                                  // weight buffers of steps s and s + D (no division in the loop)
    uint4 fa[2][NPL], fb[4][NPL];       // 16-byte fragments (eight pieces of either format)
#ifdef UC_TRACE
    // clock sums over the main loop for ONE wave (block 0, wave 0): [0] copies issued, [1] fragment reads issued .. first operands
    // there (inside step_products), [2] products issued, [3] wait + barrier, [4] steps, [5] loop total
    unsigned long long tr_copy = 0, tr_read = 0, tr_mfma = 0, tr_bar = 0, tr_t0 = __builtin_amdgcn_s_memtime();
    const bool tracing = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0;      // (a workgroup in the middle of the launch: warm caches, busy chip)
    unsigned long long tr_wait = 0, t_w = 0;
#define UC_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define UC_WB t_w = uc_wait_barrier_timed
#else
#define UC_T(var)
#define UC_WB uc_wait_barrier
#endif
    // One step's fragments and products.  A step opens behind a barrier, so all eight waves read at once: 96 KB through the CU's
    // 128-byte-per-clock LDS port, ~770 clocks.  The products therefore start as soon as the operands of the FIRST product group
    // (the low activation piece x the high weight piece: 6 of the 12 fragments) are there, and the other six reads are issued
    // behind that group's first MFMA, where they run in the matrix pipe's shadow (sched_barrier pins the order; left to itself the
    // compiler issues all twelve reads and waits for all of them in front of the first MFMA).
    // products from the smallest magnitude class to the largest; the same accumulator every 8 MFMAs
#define UC_PRODUCT(PA, PB)                                                                                      \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int n = 0; n < 4; ++n)                 \
        acc[m][n] = uc_mfma<F16>(fb[n][PB], fa[m][PA], acc[m][n]);
    auto step_products = [&](int chunk, int tap) {
        const unsigned char *abuf = sA + (chunk % (DA + 1)) * A_BYTES;
        const unsigned char *bbuf = sB + buf0 * B_BYTES;
        const int dy = p.dy[tap], dx = p.dx[tap];
        int aslot[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int px = (wave * 2 + m + 1 + dy) * UC_HW + (r + 1 + dx);
            aslot[m] = px * 2 + (h ^ ((px >> 3) & 1));
        }
        auto read_a = [&](int m, int pl) { fa[m][pl] = *reinterpret_cast<const uint4 *>(abuf + (pl * A_PLANE + aslot[m]) * 16); };
        auto read_b = [&](int n, int pl) { fb[n][pl] = *reinterpret_cast<const uint4 *>(bbuf + (pl * 256 + n * 64 + b_row) * 16); };
        if constexpr (TF) {
            const int mask = p.nmask[tap];          // (scalar: the classes this input offset feeds)
            read_a(0, 1); read_a(1, 1); read_a(0, 0); read_a(1, 0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (!((mask >> n) & 1)) continue;
                read_b(n, 0); read_b(n, 1);
#pragma unroll
                for (int m = 0; m < 2; ++m) {       // (the same product order as the general path: lo x hi, hi x lo, hi x hi)
                    acc[m][n] = uc_mfma<F16>(fb[n][0], fa[m][1], acc[m][n]);
                    acc[m][n] = uc_mfma<F16>(fb[n][1], fa[m][0], acc[m][n]);
                    acc[m][n] = uc_mfma<F16>(fb[n][0], fa[m][0], acc[m][n]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        } else if constexpr (NPL == 2) {
            read_a(0, 1); read_a(1, 1);
#pragma unroll
            for (int n = 0; n < 4; ++n) read_b(n, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);      // the wave whose operands are in registers goes first on the shared matrix pipe
            acc[0][0] = uc_mfma<F16>(fb[0][0], fa[0][1], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            read_a(0, 0); read_a(1, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) read_b(n, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 1; n < 4; ++n) acc[0][n] = uc_mfma<F16>(fb[n][0], fa[0][1], acc[0][n]);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[1][n] = uc_mfma<F16>(fb[n][0], fa[1][1], acc[1][n]);
            __builtin_amdgcn_sched_barrier(0);
#ifndef UC_TWO_PRODUCTS        // (timing experiment of DESIGN 8: how the kernel's time follows the MFMA count; never part of the product)
            UC_PRODUCT(0, 1)
#endif
            UC_PRODUCT(0, 0)
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
#pragma unroll
                for (int m = 0; m < 2; ++m) read_a(m, NPL - 1 - k);
#pragma unroll
                for (int n = 0; n < 4; ++n) read_b(n, k);
            }
            __builtin_amdgcn_s_setprio(1);
            UC_PRODUCT(2, 0) UC_PRODUCT(1, 1) UC_PRODUCT(0, 2)
            UC_PRODUCT(1, 0)
            UC_PRODUCT(0, 1)
            UC_PRODUCT(0, 0)
            __builtin_amdgcn_s_setprio(0);
        }
    };
    if constexpr (PAIR) {
        int c0 = 0, t0 = 0;                                  // (chunk, tap) of the step about to be multiplied
        int cb = (2 * SPB) / p.ntaps, tb = (2 * SPB) % p.ntaps;   // ... and of the next weight copy (2 SPB steps ahead)
        int bb = 0;                                          // weight buffer of the iteration's first step
        int a_issued = 0;                                    // highest chunk whose activations are on their way (chunk 0: prologue)
        auto next = [&](int &c, int &t) { if (++t == p.ntaps) { t = 0; ++c; } };
        auto wrapb = [](int b) { return b >= UC_NBBUF ? b - UC_NBBUF : b; };
        for (int s = 0; s < nsteps; s += SPB) {
            UC_T(t_a);
            // The copies go out one per step, BEHIND that step's products: an asynchronous copy blocks its wave at issue while the
            // CU's copy queue is full (clock trace: three weight copies back to back at the iteration's start held every wave for
            // ~830 clocks), and behind the products the wave has nothing else to do before the barrier anyway.  Their order relative
            // to the iteration's wait is what the counted wait relies on, and that does not change.
            const bool issue_a = c0 + 1 < nchunks && c0 + 1 > a_issued;
            const int ca = c0 + 1;
            if (issue_a) a_issued = ca;
            const bool issue_b = s + 2 * SPB < nsteps;       // (nsteps is a multiple of SPB: then all SPB copies exist)
            UC_T(t_b);
#pragma unroll
            for (int k = 0; k < SPB; ++k) {
                buf0 = wrapb(bb + k);
                step_products(c0, t0);
                next(c0, t0);
                if (k == 0 && issue_a) copy_a(ca, ca & 1);
                if (issue_b) { copy_b(cb, tb, wrapb(bb + 2 * SPB + k)); next(cb, tb); }
            }
            UC_T(t_c);
            // everything older than this iteration's copies has to be there: the weights of the next iteration's steps (issued an
            // iteration ago) and, when a chunk opens, its activations (issued at least two iterations ago)
            if (issue_b) { if (issue_a) UC_WB<A_PER + SPB * B_PER>(); else UC_WB<SPB * B_PER>(); }
            else { if (issue_a) UC_WB<A_PER>(); else UC_WB<0>(); }
            bb = wrapb(bb + SPB);
#ifdef UC_TRACE
            UC_T(t_d);
            tr_copy += t_b - t_a; tr_mfma += t_c - t_b; tr_bar += t_d - t_c; tr_wait += t_w - t_c;
#endif
        }
    } else
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        for (int tap = 0; tap < p.ntaps; ++tap, ++step) {
            UC_T(t_a);
            // copies: the next chunk's activations at the chunk's first tap, the weights of step + D
            const bool issue_a = tap == 0 && chunk + DA < nchunks;
            const bool issue_b = step + D < nsteps;
            UC_T(t_b);
            step_products(chunk, tap);
            // (behind the products: a copy blocks its wave at issue while the copy queue is full, see the loop above)
            if (issue_a) copy_a(chunk + DA, (chunk + DA) % (DA + 1));
            if (issue_b) copy_b(nc, nt, buf2);
            if (++nt == p.ntaps) { nt = 0; ++nc; }
            UC_T(t_c);
            // Before the barrier the weights of step + 1 must have landed (issued one step ago, before everything issued in
            // this step) and, when the next step opens a new chunk, its activations too.  Issue order inside a step is
            // activations first, weights second, so "at most B_PER outstanding" also covers the activations.
            if constexpr (D == 2) {
                // (DA == 2: the tile the next step may need was issued at least a step ago, before the weights waited for here)
                const bool need_a_now = DA == 1 && issue_a && p.ntaps == 1;
                if (issue_b) {
                    if (issue_a && !need_a_now) UC_WB<A_PER + B_PER>(); else UC_WB<B_PER>();
                } else {
                    if (issue_a && !need_a_now) UC_WB<A_PER>(); else UC_WB<0>();
                }
            } else {
                // newer than the weights of step + 1: the weight copies of steps step + 2 .. step + D that exist, and the next
                // chunk's activations while the chunk is younger than D - 1 taps (ntaps >= D: they are due much later)
                const int left = nsteps - (step + 2);
                const int nb = left < 0 ? 0 : (left > D - 1 ? D - 1 : left);
                const bool a_out = tap <= D - 2 && chunk + 1 < nchunks;
#define UC_WAIT_CASE(NB_) case NB_: if (a_out) UC_WB<NB_ * B_PER + A_PER>(); else UC_WB<NB_ * B_PER>(); break;
                switch (nb) {
                    UC_WAIT_CASE(0) UC_WAIT_CASE(1) UC_WAIT_CASE(2)
                    default: if (a_out) UC_WB<(D - 1) * B_PER + A_PER>(); else UC_WB<(D - 1) * B_PER>(); break;
                }
#undef UC_WAIT_CASE
                static_assert(D <= 4, "wait cases cover D <= 4");
            }
            buf0 = buf0 + 1 == UC_NBBUF ? 0 : buf0 + 1;
            buf2 = buf2 + 1 == UC_NBBUF ? 0 : buf2 + 1;
#ifdef UC_TRACE
            UC_T(t_d);
            tr_copy += t_b - t_a; tr_mfma += t_c - t_b; tr_bar += t_d - t_c; tr_wait += t_w - t_c;
#endif
        }
    }
#ifdef UC_TRACE
    if (tracing && p.trace && lane == 0) {
        unsigned long long *tp = p.trace + 16 * wave;
        tp[0] = tr_copy; tp[1] = tr_read; tp[2] = tr_mfma; tp[3] = tr_bar; tp[4] = (unsigned long long)nsteps;
        tp[5] = __builtin_amdgcn_s_memtime() - tr_t0; tp[6] = tr_wait; tp[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        tp[8] = tr_t0 - tr_enter;              // set-up + prologue (first tiles on their way and landed)
    }
    const unsigned long long tr_loop_end = __builtin_amdgcn_s_memtime();
#endif
#undef UC_PRODUCT

    // ---- epilogue: bias [-> ReLU -> scale, shift], split, store through LDS -----------------------------------------------------------
    // A lane holds 16 adjacent channels of one pixel per 32-channel block: ReLU and the
    // BatchNorm scale / shift (one fused multiply-add, packed two channels per instruction) take the per-channel constants as
    // 16-byte loads, v_cvt_pk_bf16_f32 rounds two adjacent channels into one word (nearest even; the remainder -- a packed subtract --
    // stays in the accumulator for the next piece), and eight channels leave as ONE 16-byte LDS write: no lane exchange.  Stored
    // straight from the registers a wave's store would touch 32-byte pieces of 32 pixel rows (the store path, not HBM, bound that
    // variant), so a tile row goes through the (now idle) LDS as a [32 pixels][128 channels] image (272-byte rows: a 16-lane
    // group's 16-byte accesses hit 16 different bank quads) and is read back four whole pixels -- 4 x 256 contiguous bytes -- per
    // instruction.  One (tile row, piece) at a time, 8.5 KB per wave.  With pool_out set (Conv2D followed by MaxPool2D(2),
    // pl.py:42-43) the 2 x 2 window of a pooled pixel is the lane's two tile rows x the neighbouring lane (one DPP max): the pooled
    // map goes out the same way first (max-then-split equals the pooled split map: the pieces are a monotone function of the value).
    // No barrier in front: every wave reads its last fragments and sees its last copies land BEFORE the final step's barrier, so a
    // wave that is past that barrier may overwrite the tile buffers.
    constexpr int EP_ROW = 272, EP_BYTES = 32 * EP_ROW;
    static_assert(UC_THREADS / 64 * EP_BYTES <= (DA + 1) * A_BYTES + UC_NBBUF * B_BYTES, "the staging images fit the tile buffers");
    int pxl = lane & 31, hf = lane >> 5;
    asm volatile("" : "+v"(pxl), "+v"(hf));           // (opaque: nothing of the epilogue's addressing is hoisted into the main loop's registers)
    unsigned char *ep = smem + wave * EP_BYTES;
    {
        const float *bip = p.bias + nblk * UC_BN + 16 * hf;
        const float *scp = p.scale + nblk * UC_BN + 16 * hf, *shp = p.shift + nblk * UC_BN + 16 * hf;
        const f32x2 inv2 = {p.acc_scale, p.acc_scale};      // fp16 pieces: accumulator -> the layer's units (a power of two)
        // accumulator + bias (fp16 pieces: accumulator x acc_scale + bias, one fused multiply-add -- the product is exact)
        auto biased = [&](float a0, float a1, const f32x2 &b) -> f32x2 {
            if constexpr (F16) return __builtin_elementwise_fma(f32x2{a0, a1}, inv2, b);
            else return f32x2{a0, a1} + b;
        };
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4 *>(bip + n * 32 + q * 4);
                const f32x2 b01 = {b4.x, b4.y}, b23 = {b4.z, b4.w};
                if (p.scale) {
                    const float4 s4 = *reinterpret_cast<const float4 *>(scp + n * 32 + q * 4), t4 = *reinterpret_cast<const float4 *>(shp + n * 32 + q * 4);
                    const f32x2 s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w}, t01 = {t4.x, t4.y}, t23 = {t4.z, t4.w};
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        f32x2 v01 = biased(acc[m][n][q * 4 + 0], acc[m][n][q * 4 + 1], b01);
                        f32x2 v23 = biased(acc[m][n][q * 4 + 2], acc[m][n][q * 4 + 3], b23);
                        v01 = __builtin_elementwise_fma(f32x2{relu_bits(v01[0]), relu_bits(v01[1])}, s01, t01);
                        v23 = __builtin_elementwise_fma(f32x2{relu_bits(v23[0]), relu_bits(v23[1])}, s23, t23);
                        if constexpr (F16) {         // (stored values are scaled: s and t carry the activation scale)
                            v01 = f32x2{uc_sat_f16(v01[0]), uc_sat_f16(v01[1])};
                            v23 = f32x2{uc_sat_f16(v23[0]), uc_sat_f16(v23[1])};
                        }
                        acc[m][n][q * 4 + 0] = v01[0]; acc[m][n][q * 4 + 1] = v01[1];
                        acc[m][n][q * 4 + 2] = v23[0]; acc[m][n][q * 4 + 3] = v23[1];
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        f32x2 v01 = biased(acc[m][n][q * 4 + 0], acc[m][n][q * 4 + 1], b01);     // (fp16 pieces, bias only: acc_scale and the bias carry the activation scale)
                        f32x2 v23 = biased(acc[m][n][q * 4 + 2], acc[m][n][q * 4 + 3], b23);
                        if constexpr (F16) {
                            v01 = f32x2{uc_sat_f16(v01[0]), uc_sat_f16(v01[1])};
                            v23 = f32x2{uc_sat_f16(v23[0]), uc_sat_f16(v23[1])};
                        }
                        acc[m][n][q * 4 + 0] = v01[0]; acc[m][n][q * 4 + 1] = v01[1];
                        acc[m][n][q * 4 + 2] = v23[0]; acc[m][n][q * 4 + 3] = v23[1];
                    }
                }
            }
            asm volatile("" ::: "memory");       // (one block's constants at a time: the loads are not all hoisted in front)
        }
    }
    if (p.head_out) {
        // The network's head (pl.py:69: Conv2D(2, 1) + softmax over the two classes) on the float32 values in the registers: a lane
        // holds 64 of its two pixels' 128 channels, the other 64 sit in lane ^ 32.  The layer's own output has no other reader and
        // is not stored (2.1 GB less to write and to read back at 2048^2, and the head sees unsplit float32 values).
        float z[2][2] = {{0.f, 0.f}, {0.f, 0.f}};       // [tile row m][class]
        const float *hw0 = p.head_w + 16 * hf, *hw1 = p.head_w + UC_BN + 16 * hf;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a4 = *reinterpret_cast<const float4 *>(hw0 + n * 32 + q * 4), b4 = *reinterpret_cast<const float4 *>(hw1 + n * 32 + q * 4);
                const float wa[4] = {a4.x, a4.y, a4.z, a4.w}, wb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        z[m][0] = __builtin_fmaf(acc[m][n][q * 4 + j], wa[j], z[m][0]);
                        z[m][1] = __builtin_fmaf(acc[m][n][q * 4 + j], wb[j], z[m][1]);
                    }
            }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c) z[m][c] += __shfl_xor(z[m][c], 32, 64);
        const float z0 = (hf ? z[1][0] : z[0][0]) + p.head_b[0], z1 = (hf ? z[1][1] : z[0][1]) + p.head_b[1];     // half-wave hf: tile row hf
        const float zm = z0 > z1 ? z0 : z1;
        const float e0 = __expf(z0 - zm), e1 = __expf(z1 - zm);
        const float es = e0 + e1;
        const long o = (long)(ty0 + wave * 2 + hf) * p.W + tx0 + pxl;
        p.head_out[o] = e0 / es;
        p.head_out[(long)p.H * p.W + o] = e1 / es;
        return;
    }
    auto piece_word = [](float &a, float &b, bool more) -> unsigned { return uc_piece_word<F16>(a, b, more); };
    const int rd_row = lane >> 4, rd_col = (lane & 15) * 16;       // read-back: 16 lanes = one pixel's 256 bytes
    const long out_plane = (long)p.outH * p.outW * (TF ? p.cout / 4 : p.cout);
    if (p.pool_out) {
        const long pool_plane = (long)(p.outH / 2) * (p.outW / 2) * p.cout;
        const bool oddp = pxl & 1;
        float q8[4][8];            // the lane's half of the pooled pixel's channels: even lanes 0..7, odd lanes 8..15 of every block
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = fmax_pair(fmax_raw(acc[0][n][i], acc[1][n][i]));
                if (i < 8) q8[n][i] = v; else if (oddp) q8[n][i - 8] = v;
            }
        uint16_t *prow = p.pool_out + ((long)(ty0 / 2 + wave) * (p.outW / 2) + tx0 / 2) * p.cout + nblk * UC_BN + (lane & 15) * 8;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                uint4 w;
                w.x = piece_word(q8[n][0], q8[n][1], pl + 1 < NPL);
                w.y = piece_word(q8[n][2], q8[n][3], pl + 1 < NPL);
                w.z = piece_word(q8[n][4], q8[n][5], pl + 1 < NPL);
                w.w = piece_word(q8[n][6], q8[n][7], pl + 1 < NPL);
                *reinterpret_cast<uint4 *>(ep + (pxl >> 1) * EP_ROW + (n * 32 + 16 * hf + (oddp ? 8 : 0)) * 2) = w;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + rd_row;
                *reinterpret_cast<uint4 *>(prow + pl * pool_plane + (long)row * p.cout) = *reinterpret_cast<const uint4 *>(ep + row * EP_ROW + rd_col);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int y = ty0 + wave * 2 + m;
        // TF: the 16-byte piece j = lane & 15 of a staged pixel is class j / 4, channels 8 (j % 4) .. of the workgroup's 32: output pixel
        // (2 y + class / 2, 2 x + class % 2), pixel stride = the real channel count
        const int cls = (lane & 15) >> 2;
        const long pix_stride = TF ? p.cout / 4 : p.cout;
        uint16_t *orow = TF ? p.out + ((long)(2 * y + (cls >> 1)) * p.outW + (2 * (long)tx0 + (cls & 1))) * pix_stride + nblk * 32 + (lane & 3) * 8
                            : p.out + ((long)(y * p.sy + p.oy) * p.outW + ((long)tx0 * p.sx + p.ox)) * p.cout + nblk * UC_BN + (lane & 15) * 8;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = acc[m][n][q * 8 + j];      // (vector elements do not bind to references)
                    uint4 w;
                    w.x = piece_word(v[0], v[1], pl + 1 < NPL);
                    w.y = piece_word(v[2], v[3], pl + 1 < NPL);
                    w.z = piece_word(v[4], v[5], pl + 1 < NPL);
                    w.w = piece_word(v[6], v[7], pl + 1 < NPL);
                    if (pl + 1 < NPL) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[m][n][q * 8 + j] = v[j];
                    }
                    *reinterpret_cast<uint4 *>(ep + pxl * EP_ROW + (n * 32 + 16 * hf + 8 * q) * 2) = w;
                }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + rd_row;
                *reinterpret_cast<uint4 *>(orow + pl * out_plane + (long)row * p.sx * pix_stride) = *reinterpret_cast<const uint4 *>(ep + row * EP_ROW + rd_col);
            }
        }
    }
#ifdef UC_TRACE
    if (tracing && p.trace && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stores accepted: what a following workgroup would wait for)
        p.trace[16 * wave + 9] = __builtin_amdgcn_s_memtime() - tr_loop_end;
    }
#endif
#endif
}

// ---- first layer: Conv2D(2 -> 128, 3x3) on the float32 (2, H, W) network input --------------------------------------------------
// K = 18: nothing for the matrix cores -- the layer is its 2.1 GB of output.  Exact float32 FMAs; a thread owns 4 adjacent output
// channels, keeps their 18 x 4 weights in registers and walks a run of FIRST_RUN consecutive pixels of one row with the 3 x 3 x 2
// input window sliding through registers (six loads and no address arithmetic per pixel: with per-pixel tap addressing the kernel
// was bound by its own index arithmetic, 210 vector instructions per pixel and thread, 0.8 ms against 0.45 ms for the stores
// alone).  Thirty-two threads share a pixel, so the split of two adjacent channels is one v_cvt_pk_bf16_f32 (no lane exchange) and
// a pixel's 256 bytes per piece leave in one piece.
constexpr int FIRST_RUN = 32;       // consecutive pixels per 32-thread slot; a 256-thread block covers 8 runs = 256 pixels of a row
constexpr int FIRST_PIX = 8 * FIRST_RUN;
template <int NPL, bool F16 = false>
__global__ void __launch_bounds__(256) k_unet_conv_first(const float *__restrict__ in, int H, int W, const float *__restrict__ wgt /* [9][2][128] */,
                                                         const float *__restrict__ bias, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, uint16_t *__restrict__ out)
{
    const int cg = threadIdx.x & 31, slot = threadIdx.x >> 5;
    float4 w4[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) w4[k] = *reinterpret_cast<const float4 *>(wgt + k * 128 + cg * 4);
    const float4 bb = *reinterpret_cast<const float4 *>(bias + cg * 4);
    const float4 ss = *reinterpret_cast<const float4 *>(scale + cg * 4);
    const float4 tt = *reinterpret_cast<const float4 *>(shift + cg * 4);
    const long plane = (long)H * W * 128, chan = (long)H * W;
    const long pix0 = (long)blockIdx.x * FIRST_PIX + slot * FIRST_RUN;        // (W % FIRST_RUN == 0: a run stays inside one row)
    const int y = (int)(pix0 / W), x0 = (int)(pix0 - (long)y * W);
    // the three input rows (clamped addresses, zero where the row lies outside the image: 'same' padding)
    const float *rowp[3];
    bool rowin[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        rowin[dy] = yy >= 0 && yy < H;
        rowp[dy] = in + (long)min(max(yy, 0), H - 1) * W;
    }
    auto column = [&](int xx, float (&c)[3][2]) {          // input column xx of the window: [row][channel]
        const bool xin = xx >= 0 && xx < W;
        const int xc = min(max(xx, 0), W - 1);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float a = rowp[dy][xc], b = rowp[dy][chan + xc];
            c[dy][0] = xin && rowin[dy] ? a : 0.f;
            c[dy][1] = xin && rowin[dy] ? b : 0.f;
        }
    };
    float win[3][3][2];          // [column slot][row][channel]; slot (it + k) % 3 holds column x - 1 + k
    column(x0 - 1, win[0]);
    column(x0, win[1]);
    uint16_t *dst = out + pix0 * 128 + cg * 4;
    for (int it0 = 0; it0 < FIRST_RUN; it0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {        // (unrolled by the window's period: every slot index is a compile-time constant)
            const int it = it0 + u;
            if (it >= FIRST_RUN) break;
            column(x0 + it + 1, win[(u + 2) % 3]);
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};       // packed FMAs: two channels per instruction
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v0 = win[(u + t % 3) % 3][t / 3][0], v1 = win[(u + t % 3) % 3][t / 3][1];
                a01 = __builtin_elementwise_fma(f32x2{v0, v0}, f32x2{w4[2 * t].x, w4[2 * t].y}, a01);
                a23 = __builtin_elementwise_fma(f32x2{v0, v0}, f32x2{w4[2 * t].z, w4[2 * t].w}, a23);
                a01 = __builtin_elementwise_fma(f32x2{v1, v1}, f32x2{w4[2 * t + 1].x, w4[2 * t + 1].y}, a01);
                a23 = __builtin_elementwise_fma(f32x2{v1, v1}, f32x2{w4[2 * t + 1].z, w4[2 * t + 1].w}, a23);
            }
            float a[4] = {a01[0] + bb.x, a01[1] + bb.y, a23[0] + bb.z, a23[1] + bb.w};
            const float sv[4] = {ss.x, ss.y, ss.z, ss.w}, tv[4] = {tt.x, tt.y, tt.z, tt.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = a[j] > 0.f ? a[j] : 0.f;
                a[j] = r * sv[j] + tv[j];
                if constexpr (F16) a[j] = uc_sat_f16(a[j]);      // (s and t carry the activation scale)
            }
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                unsigned wd[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) wd[q] = uc_piece_word<F16>(a[2 * q], a[2 * q + 1], pl + 1 < NPL);
                *reinterpret_cast<uint2 *>(dst + (long)it * 128 + pl * plane) = make_uint2(wd[0], wd[1]);
            }
        }
    }
}

// value of split element e (0..7) of the 16-byte pieces pc[0..NPL)
template <int NPL, bool F16 = false>
__device__ __forceinline__ float split_value(const uint4 *pc, int e)
{
    float v = 0.f;
#pragma unroll
    for (int pl = NPL - 1; pl >= 0; --pl) {     // smallest piece first: the sum is exact either way (<= 24 significant bits)
        const unsigned w = e < 2 ? pc[pl].x : (e < 4 ? pc[pl].y : (e < 6 ? pc[pl].z : pc[pl].w));
        v += uc_piece_value<F16>((e & 1) ? (w >> 16) : (w & 0xffffu));
    }
    return v;
}

// ---- MaxPool2D(2) on split planes: the winner's pieces are copied (the pieces of a value are a function of the value) ---------
template <int NPL, bool F16 = false>
__global__ void __launch_bounds__(256) k_unet_pool2(const uint16_t *__restrict__ in, int H, int W, int C, uint16_t *__restrict__ out)
{
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Ho * Wo * C8) return;
    const int c8 = (int)(i % C8);
    const long op = i / C8;
    const int xo = (int)(op % Wo), yo = (int)(op / Wo);
    const long in_plane = (long)H * W * C, out_plane = (long)Ho * Wo * C;
    uint4 pc[4][NPL];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            pc[k][pl] = *reinterpret_cast<const uint4 *>(in + pl * in_plane + ((long)(2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C + c8 * 8);
    unsigned res[NPL][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int best = 0;
        float bv = split_value<NPL, F16>(pc[0], e);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float v = split_value<NPL, F16>(pc[k], e);
            if (v > bv) { bv = v; best = k; }
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const uint4 q = best == 0 ? pc[0][pl] : (best == 1 ? pc[1][pl] : (best == 2 ? pc[2][pl] : pc[3][pl]));
            const unsigned w = e < 2 ? q.x : (e < 4 ? q.y : (e < 6 ? q.z : q.w));
            const unsigned hb = (e & 1) ? (w >> 16) : (w & 0xffffu);
            if (e & 1) res[pl][e >> 1] |= hb << 16; else res[pl][e >> 1] = hb;
        }
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<uint4 *>(out + pl * out_plane + op * C + c8 * 8) = make_uint4(res[pl][0], res[pl][1], res[pl][2], res[pl][3]);
}

// ---- head: Conv2D(128 -> 2, 1x1) + softmax over the two classes (pl.py:69), float32 out (2, H, W) -------------------------------
// eight lanes per pixel, 16 channels each; logits != 0: the pre-softmax values (the bench's head calibration)
template <int NPL, bool F16 = false>
__global__ void __launch_bounds__(256) k_unet_head(const uint16_t *__restrict__ in, long npix, const float *__restrict__ wgt /* [2][128] */,
                                                   const float *__restrict__ bias, float *__restrict__ out, int logits)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pix = t >> 3;
    const int part = (int)(t & 7);
    const long plane = npix * 128;
    float z0 = 0.f, z1 = 0.f;
    if (pix < npix) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 pc[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) pc[pl] = *reinterpret_cast<const uint4 *>(in + pl * plane + pix * 128 + part * 16 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = split_value<NPL, F16>(pc, e);
                const int c = part * 16 + g * 8 + e;
                z0 = __builtin_fmaf(v, wgt[c], z0);
                z1 = __builtin_fmaf(v, wgt[128 + c], z1);
            }
        }
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) { z0 += __shfl_xor(z0, d, 64); z1 += __shfl_xor(z1, d, 64); }
    if (pix < npix && part == 0) {
        z0 += bias[0]; z1 += bias[1];
        if (logits) { out[pix] = z0; out[npix + pix] = z1; return; }
        const float m = z0 > z1 ? z0 : z1;
        const float e0 = __expf(z0 - m), e1 = __expf(z1 - m);
        const float s = e0 + e1;
        out[pix] = e0 / s;
        out[npix + pix] = e1 / s;
    }
}

}  // namespace tip
