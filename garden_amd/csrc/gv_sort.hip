// gv_sort.hip — gv_sort: sortMeshes (source/system/render/mesh.cpp:265-328) on the device.
#include "gv_device.hpp"

namespace gv {

// ------------------------------------------------------------------------------------------------
// sortMeshes (mesh.cpp:265-328): order the compact records by distanceSq — ascending for unsorted buffers
// (front to back, operator< at render/mesh.hpp:196), descending for the sorted / translucent ones (:204).
// Stable LSD radix sort on a 32-bit order-preserving key, 8 bits per pass; ties keep ascending slot order
// (std::sort in the reference is unstable, so any tie order is within its contract).
// ------------------------------------------------------------------------------------------------
// Large pools: a single-pass-per-digit ("onesweep") LSD radix sort, 5 launches instead of 14:
//   sort_prepare_kernel   one pass over distanceSq: the GLOBAL histograms of all four 8-bit digits at once (the digit
//                         totals do not depend on the order of the keys, so they need not wait for the passes), and
//                         the housekeeping of this sort's look-back state;
//   onesweep_kernel x 4   per digit: a workgroup ranks its tile of 4096 keys (stable), publishes the tile's digit
//                         counts and obtains the counts of all tiles before it by DECOUPLED LOOK-BACK (no histogram and
//                         scan launches between the passes), reorders the tile by digit in LDS and writes each digit's
//                         run to its final place. Pass 0 builds the keys from distanceSq itself; the LAST pass does not
//                         write (key, index) pairs at all but gathers the 56-byte records straight to their sorted
//                         positions (the separate key and gather launches of the 14-launch form are gone).
// 2.1 M records move 8 + 16 + 16 + 16 + 8 B of pairs and 112 B of records each = 176 B; the working set (17 MB of pairs
// at 2.1 M) stays in L2 / Infinity Cache, so the passes are latency- rather than HBM-bound: what matters is launches,
// barriers and atomic conflicts (keys of one frame share their top bytes: every counter update is aggregated per wave
// by digit matching first).
#ifdef GV_SORT_TRACE  // dev tool only (tools/onesweep_probe.hip): wall-clock stamps per tile and phase
__device__ unsigned long long gv_sort_trace[4][8192][8];
#define GV_TRACE(k)                                                    \
    if (threadIdx.x == 0 && tile < 8192)                               \
        gv_sort_trace[a.pass][tile][k] = wall_clock64();
#define GV_TRACE_AFTER_LOADS(slot)                                     \
    __builtin_amdgcn_s_waitcnt(0);                                     \
    __builtin_amdgcn_sched_barrier(0);                                 \
    if (threadIdx.x == 0 && tile < 8192)                               \
        gv_sort_trace[a.pass][tile][slot] = wall_clock64();            \
    __builtin_amdgcn_sched_barrier(0);
#else
#define GV_TRACE(k)
#define GV_TRACE_AFTER_LOADS(slot)
#endif
constexpr uint32_t kSortTile = 4096;               // keys per workgroup per pass
constexpr uint32_t kSortRounds = kSortTile / 256;  // keys per lane
#ifndef GV_SORT_LOOK_WINDOW
#define GV_SORT_LOOK_WINDOW 16
#endif
constexpr uint32_t kLookWindow = GV_SORT_LOOK_WINDOW;  // predecessors inspected per look-back step
constexpr uint32_t kFlagAggregate = 1u << 30, kFlagPrefix = 2u << 30, kFlagMask = 3u << 30, kCountMask = ~kFlagMask;

__device__ __forceinline__ uint32_t order_key(float d, uint32_t descending)
{
    const uint32_t u = __float_as_uint(d);
    const uint32_t k = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending float order == ascending key order
    return descending ? ~k : k;
}

// lanes of the wave whose `digit` equals this lane's (among `valid` lanes): 8 ballots
__device__ __forceinline__ unsigned long long match_digit(uint32_t digit, bool valid)
{
    unsigned long long peer = __ballot(valid);
#pragma unroll
    for (uint32_t b = 0; b < 8; b++) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peer &= bit ? m : ~m;
    }
    return peer;
}

// exclusive scan of one value per thread over a 256-thread workgroup; every thread also gets the grand total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sum /* LDS [4] */, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d)
            incl += up;
    }
    __syncthreads();  // wave_sum may still be read from a previous scan
    if (lane == 63)
        wave_sum[wave] = incl;
    __syncthreads();
    uint32_t wave_prefix = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        wave_prefix += w < wave ? wave_sum[w] : 0u;
        all += wave_sum[w];
    }
    *total = all;
    return wave_prefix + incl - v;
}

// LDS operations of one wave execute in issue order; the fences keep the compiler from reordering them
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct SortState {
    uint32_t* ghist;         // this sort's global digit histograms [4][256]   (zero on entry)
    uint32_t* tile_counter;  // this sort's dynamic tile counters [4]          (zero on entry)
    uint32_t* ghist_next;    // the other parity's set, zeroed here for the next sort
    uint32_t* tile_counter_next;
    uint32_t* status;        // look-back words [4][tile_stride][256]
    uint32_t tile_stride;
};

// Global histograms of the four digits + housekeeping: one workgroup per tile of 4096 keys (surplus workgroups of the
// capacity-sized grid leave at once), 16 keys per lane loaded up front, LDS counters, then one global atomic per
// non-empty counter. Keys of one frame share their high bytes, so a wave whose 64 keys agree on a digit adds 64 with one
// lane instead of queueing 64 same-address LDS atomics.
__global__ __launch_bounds__(256) void sort_prepare_kernel(const float* __restrict__ dist, const uint32_t* __restrict__ count,
                                                           uint32_t capacity, uint32_t descending, SortState st)
{
    __shared__ uint32_t bins[4][256];
    const uint32_t n = min(*count, capacity);
    const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
    if (blockIdx.x == 0) {  // the other parity's counters, for the next sort (nobody reads them during this one)
        for (uint32_t k = threadIdx.x; k < 4 * 256; k += 256)
            st.ghist_next[k] = 0;
        if (threadIdx.x < 4)
            st.tile_counter_next[threadIdx.x] = 0;
    }
    if (blockIdx.x >= tiles)
        return;
#pragma unroll
    for (uint32_t p = 0; p < 4; p++) {
        bins[p][threadIdx.x] = 0;
        st.status[((size_t)p * st.tile_stride + blockIdx.x) * 256 + threadIdx.x] = 0;  // this tile's look-back words
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t base = blockIdx.x * kSortTile + (threadIdx.x >> 6) * (kSortTile / 4);
    uint32_t key[kSortRounds];
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t j = base + r * 64 + lane;
        key[r] = j < n ? order_key(dist[j], descending) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t j = base + r * 64 + lane;
        const bool valid = j < n;
        const unsigned long long live = __ballot(valid);
        if (live == 0ull)
            break;  // wave-uniform: the rest of this wave's keys lie beyond n
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)key[r]);  // lane 0 is valid whenever any lane is
        const uint32_t diff = key[r] ^ first;
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            const uint32_t d = (key[r] >> (8 * p)) & 255u;
            const bool same = __ballot(valid && ((diff >> (8 * p)) & 255u) != 0u) == 0ull;  // every live key has lane 0's digit
            if (same) {
                if (lane == 0)
                    atomicAdd(&bins[p][d], (uint32_t)__popcll(live));
            } else if (valid) {
                atomicAdd(&bins[p][d], 1u);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t p = 0; p < 4; p++) {
        const uint32_t c = bins[p][threadIdx.x];
        if (c)
            atomicAdd(&st.ghist[p * 256 + threadIdx.x], c);
    }
}

struct SortPassArgs {
    const uint32_t* count;
    uint32_t capacity, descending, pass;
    uint32_t static_tiles;      // GV_DEBUG_SORT_STATIC_TILES (measurement only)
    const float* dist_in;       // FIRST: keys are built from these
    const uint32_t* keys_in;    // !FIRST
    const uint32_t* vals_in;    // !FIRST
    uint32_t* keys_out;         // !LAST
    uint32_t* vals_out;         // !LAST
    const uint32_t* idx_in;     // LAST: the records, gathered to their sorted positions
    const float* model_in;
    uint32_t* idx_out;
    float* model_out;
    float* dist_out;
    SortState st;
};

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void onesweep_kernel(const SortPassArgs a)
{
    __shared__ uint32_t wcount[4][256];   // per-wave digit counts of the tile, then per-wave digit bases
    __shared__ uint32_t tile_excl[256];   // exclusive scan of the tile's digit counts (tile-local sorted order)
    __shared__ uint32_t dst_base[256];    // global position of the tile's first key of each digit
    __shared__ uint32_t skey[kSortTile];  // the tile reordered by digit
    __shared__ uint32_t sval[kSortTile];
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t tile_id;
    const uint32_t n = min(*a.count, a.capacity);
    const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
    if (blockIdx.x >= tiles)
        return;  // the grid is sized for the capacity, the count lives on the device: surplus workgroups leave at once
    if (threadIdx.x == 0)
        tile_id = a.static_tiles ? blockIdx.x                  // debug A/B only: relies on in-order workgroup dispatch
                                 : atomicAdd(&a.st.tile_counter[a.pass], 1u);  // tiles are taken in order of arrival: a workgroup
                                                              // only ever waits for tiles whose workgroups are already running
#pragma unroll
    for (uint32_t w = 0; w < 4; w++)
        wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = tile_id;  // < tiles: exactly `tiles` workgroups take one each
    GV_TRACE(0)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t shift = a.pass * 8;
    // ---- A: each wave ranks its 1024 consecutive keys, 64 per round: wave-private counters, no workgroup barrier ----
    const uint32_t wave_base = tile * kSortTile + wave * (kSortTile / 4);
    uint32_t key[kSortRounds], val[kSortRounds], local[kSortRounds];
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {  // all loads first
        const uint32_t j = wave_base + r * 64 + lane;
        const bool valid = j < n;
        if (FIRST) {
            key[r] = valid ? order_key(a.dist_in[j], a.descending) : 0xFFFFFFFFu;
            val[r] = j;
        } else {
            key[r] = valid ? a.keys_in[j] : 0xFFFFFFFFu;
            val[r] = valid ? a.vals_in[j] : 0u;
        }
    }
    // Ranking: per round, the lanes sharing my digit (a wave whose live keys all agree — the high bytes of one frame's
    // distances — knows without asking; otherwise 8 ballots), my rank among them, and the wave's running count of that
    // digit from a wave-private LDS table (the LDS executes a wave's instructions in order: round r sees exactly the
    // counts of rounds < r, so the ranking is stable). Two other forms were built and measured no faster on the box
    // (profiles/r02_sort_probe.txt): LDS lane-mask tables instead of the ballots (fewer instructions, more LDS round
    // trips), and all sixteen rounds' counts taken with returning LDS atomics in flight together (needs > 128 VGPRs).
    GV_TRACE_AFTER_LOADS(6)
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t j = wave_base + r * 64 + lane;
        const bool valid = j < n;
        const uint32_t d = (key[r] >> shift) & 255u;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);  // lane 0 is valid whenever any lane is
        const unsigned long long peer = __ballot(valid && d != d0) != 0ull ? match_digit(d, valid) : __ballot(valid);
        const uint32_t below = (uint32_t)__popcll(peer & ((1ull << lane) - 1ull));
        const uint32_t prior = wcount[wave][d];
        wave_lds_order();
        if (valid && below == 0)
            wcount[wave][d] = prior + (uint32_t)__popcll(peer);
        wave_lds_order();
        local[r] = prior + below;  // rank among this wave's keys of digit d
    }
    __syncthreads();
    GV_TRACE(1)
    // ---- B: one thread per digit: tile counts -> look-back -> this tile's global base per digit ----
    const uint32_t d = threadIdx.x;
    const uint32_t c0 = wcount[0][d], c1 = wcount[1][d], c2 = wcount[2][d], c3 = wcount[3][d];
    const uint32_t tile_count = c0 + c1 + c2 + c3;
    uint32_t* status = a.st.status + ((size_t)a.pass * a.st.tile_stride + tile) * 256;
    // Decoupled look-back, a window at a time: all kLookWindow loads of a step are independent (one memory round trip
    // for up to 16 predecessors instead of one per predecessor — with every tile resident at once a serial walk grows
    // like sqrt(2 * tiles) round trips, which was most of a pass), consumed nearest first up to the first inclusive
    // prefix; words not published yet are polled again.
    uint32_t before = 0;  // keys of digit d in the tiles before this one
    if (tile == 0) {
        __hip_atomic_store(&status[d], kFlagPrefix | tile_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(&status[d], kFlagAggregate | tile_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t* first_tile = a.st.status + (size_t)a.pass * a.st.tile_stride * 256 + d;
        int32_t p = (int32_t)tile - 1;  // nearest predecessor not consumed yet
        bool done = false;
        while (!done) {
            uint32_t v[kLookWindow];
#pragma unroll
            for (int32_t i = 0; i < (int32_t)kLookWindow; i++)
                v[i] = p - i >= 0 ? __hip_atomic_load(first_tile + (size_t)(p - i) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : kFlagPrefix;  // in front of tile 0: nothing
            bool open = true;
            int32_t consumed = 0;
#pragma unroll
            for (int32_t i = 0; i < (int32_t)kLookWindow; i++) {
                const uint32_t flag = v[i] & kFlagMask;
                open = open && flag != 0;
                if (open) {
                    before += v[i] & kCountMask;
                    consumed++;
                    if (flag == kFlagPrefix) {
                        done = true;
                        open = false;
                    }
                }
            }
            p -= consumed;
            if (consumed == 0)
                __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(&status[d], kFlagPrefix | (before + tile_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    GV_TRACE(2)
    uint32_t total;
    const uint32_t gbase = block_exclusive_scan(a.st.ghist[a.pass * 256 + d], wave_sum, &total);  // keys of lower digits, all tiles
    const uint32_t texcl = block_exclusive_scan(tile_count, wave_sum, &total);                   // ... in this tile
    tile_excl[d] = texcl;
    dst_base[d] = gbase + before;
    wcount[0][d] = texcl;  // per-wave bases in the tile-local sorted order
    wcount[1][d] = texcl + c0;
    wcount[2][d] = texcl + c0 + c1;
    wcount[3][d] = texcl + c0 + c1 + c2;
    __syncthreads();
    GV_TRACE(3)
    // ---- C: reorder the tile by digit in LDS, then write every digit's run to its place ----
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t j = wave_base + r * 64 + lane;
        if (j < n) {
            const uint32_t lpos = wcount[wave][(key[r] >> shift) & 255u] + local[r];
            skey[lpos] = key[r];
            sval[lpos] = val[r];
        }
    }
    __syncthreads();
    GV_TRACE(4)
    const uint32_t live = min(kSortTile, n - tile * kSortTile);
    if (!LAST) {
#pragma unroll 4
        for (uint32_t t = threadIdx.x; t < live; t += 256) {
            const uint32_t k = skey[t], v = sval[t];
            const uint32_t dd = (k >> shift) & 255u;
            const uint32_t pos = dst_base[dd] + (t - tile_excl[dd]);
            a.keys_out[pos] = k;
            a.vals_out[pos] = v;
        }
        GV_TRACE(5)
        return;
    }
    // LAST: the records go straight to their sorted positions. distanceSq is the key itself (order_key is a bijection:
    // no gather), the pool slot is a 4-byte gather from an array that fits the caches, the 48-byte model is gathered by
    // three lanes per record (one float4 each), so that the stores of a run of records are whole contiguous rows.
    for (uint32_t t = threadIdx.x; t < live; t += 256) {
        const uint32_t k = skey[t], v = sval[t];  // v = the record's index before the sort
        const uint32_t dd = (k >> shift) & 255u;
        const uint32_t pos = dst_base[dd] + (t - tile_excl[dd]);
        const uint32_t u = a.descending ? ~k : k;
        a.dist_out[pos] = __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        a.idx_out[pos] = a.idx_in[v];
        skey[t] = pos;  // only this thread reads skey[t] in this loop
    }
    __syncthreads();
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a.model_in);
    float4* __restrict__ dst = reinterpret_cast<float4*>(a.model_out);
#pragma unroll 4
    for (uint32_t q = threadIdx.x; q < live * 3u; q += 256) {
        const uint32_t t = q / 3u, part = q - t * 3u;
        dst[(size_t)skey[t] * 3 + part] = src[(size_t)sval[t] * 3 + part];
    }
    GV_TRACE(5)
}

// Small pools (up to kSmallSort records possible): ONE launch instead of fourteen — a tick of an engine-sized scene
// (10^4 entities) is launch-bound. Rank sort: the position of record i is the number of records that order before
// it, (key, emission index) compared as a pair, so the result is the stable order the radix passes produce. A
// workgroup owns 64 records; its four waves each count over a quarter of the keys (staged in LDS, read as uniform
// 16-byte broadcasts) and the partial counts meet in LDS. n^2 / 4 comparisons per wave, all CUs busy, no
// inter-workgroup step: ~6 us at 2 k records where a one-workgroup bitonic network took 42 us.
constexpr uint32_t kSmallSort = kSmallSortMaxSlots;

// records of [jlo, jhi) (multiples of 4) that order before record i with key ki. WHERE: 0 = every j is below the
// workgroup's records (ties count), 2 = every j is above them (ties do not), 1 = overlapping (compare the pair)
template <int WHERE>
__device__ __forceinline__ uint32_t count_before(const uint32_t* key, uint32_t jlo, uint32_t jhi, uint32_t ki, uint32_t i)
{
    uint32_t before = 0;
#pragma unroll 4
    for (uint32_t j = jlo; j < jhi; j += 4) {
        const uint4 k = *reinterpret_cast<const uint4*>(key + j);  // uniform address: one LDS broadcast per 4 keys
        if (WHERE == 0)
            before += (k.x <= ki) + (k.y <= ki) + (k.z <= ki) + (k.w <= ki);
        else if (WHERE == 2)
            before += (k.x < ki) + (k.y < ki) + (k.z < ki) + (k.w < ki);
        else
            before += ((k.x < ki) | ((k.x == ki) & (j < i))) + ((k.y < ki) | ((k.y == ki) & (j + 1 < i))) +
                      ((k.z < ki) | ((k.z == ki) & (j + 2 < i))) + ((k.w < ki) | ((k.w == ki) & (j + 3 < i)));
    }
    return before;
}

template <class Entry>
__device__ __forceinline__ void sort_small_block(const Entry& b, uint32_t capacity, uint32_t descending, uint32_t block)
{
    extern __shared__ uint32_t key[];  // order-preserving keys of all n records, padded to a multiple of 4
    __shared__ uint32_t partial[4][64];
    const uint32_t n = min(*b.count, capacity);
    const uint32_t i0 = block * 64;
    if (i0 >= n)
        return;
    const uint32_t n4 = (n + 3u) & ~3u;
    for (uint32_t j = threadIdx.x; j < n4; j += 256) {
        uint32_t k = 0xFFFFFFFFu;  // padding: never counted (its index is beyond every record's)
        if (j < n) {
            const uint32_t u = __float_as_uint(b.dist_in[j]);
            k = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
            if (descending)
                k = ~k;
        }
        key[j] = k;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t i = i0 + lane;
    const uint32_t ki = key[min(i, n4 - 1)];
    const uint32_t per = ((n4 >> 2) + 3u) & ~3u;  // keys per wave, a multiple of 4
    const uint32_t jlo = min(part * per, n4), jhi = min(jlo + per, n4);
    const uint32_t own_lo = min(max(i0 & ~3u, jlo), jhi), own_hi = min(max((i0 + 64u + 3u) & ~3u, jlo), jhi);
    const uint32_t before = count_before<0>(key, jlo, own_lo, ki, i) + count_before<1>(key, own_lo, own_hi, ki, i) +
                            count_before<2>(key, own_hi, jhi, ki, i);
    partial[part][lane] = before;
    __syncthreads();
    if (part != 0 || i >= n)
        return;
    const uint32_t rank = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
    b.idx_out[rank] = b.idx_in[i];
    b.dist_out[rank] = b.dist_in[i];
    const float4* sm = reinterpret_cast<const float4*>(b.model_in + (size_t)i * 12);
    float4* dm = reinterpret_cast<float4*>(b.model_out + (size_t)rank * 12);
    const float4 m0 = sm[0], m1 = sm[1], m2 = sm[2];
    dm[0] = m0;
    dm[1] = m1;
    dm[2] = m2;
}

__global__ __launch_bounds__(256) void sort_small_kernel(const SortBuffers b, uint32_t capacity, uint32_t descending)
{
    sort_small_block(b, capacity, descending, blockIdx.x);
}

// several views of one small pool (main camera + shadow passes) in one launch: blockIdx.y picks the view
__global__ __launch_bounds__(256) void sort_small_batch_kernel(const SortBatch batch)
{
    const SmallSortEntry& e = batch.view[blockIdx.y];
    sort_small_block(e, e.capacity, e.descending, blockIdx.x);
}

hipError_t launch_sort(const SortBuffers& b, uint32_t capacity, bool descending, hipStream_t stream)
{
    if (capacity == 0)
        return hipSuccess;
    if (capacity <= kSmallSort) {
        const uint32_t lds = ((capacity + 3u) & ~3u) * 4;  // 64 KB of keys at the limit, beside the 1 KB of partial counts
        static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_small_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
        if (raised != hipSuccess)
            return raised;
        hipLaunchKernelGGL(sort_small_kernel, dim3((capacity + 63) / 64), dim3(256), lds, stream, b, capacity, descending ? 1u : 0u);
        return hipGetLastError();
    }
    const uint32_t tiles = (capacity + kSortTile - 1) / kSortTile;  // at full capacity; the live count is on the device
    SortState st;
    st.ghist = b.ghist[b.parity];
    st.tile_counter = b.tile_counter[b.parity];
    st.ghist_next = b.ghist[b.parity ^ 1u];
    st.tile_counter_next = b.tile_counter[b.parity ^ 1u];
    st.status = b.status;
    st.tile_stride = tiles;
    hipLaunchKernelGGL(sort_prepare_kernel, dim3(tiles), dim3(256), 0, stream, b.dist_in, b.count, capacity,
                       descending ? 1u : 0u, st);
    SortPassArgs a{};
    a.count = b.count;
    a.capacity = capacity;
    a.descending = descending ? 1u : 0u;
    static const uint32_t static_tiles = getenv("GV_DEBUG_SORT_STATIC_TILES") ? 1u : 0u;
    a.static_tiles = static_tiles;
    a.dist_in = b.dist_in;
    a.idx_in = b.idx_in;
    a.model_in = b.model_in;
    a.idx_out = b.idx_out;
    a.model_out = b.model_out;
    a.dist_out = b.dist_out;
    a.st = st;
    for (uint32_t pass = 0; pass < 4; pass++) {
        const uint32_t src = (pass & 1u) ^ 1u, dst = pass & 1u;  // pass 0 writes set 0, pass 1 set 1, ...
        a.pass = pass;
        a.keys_in = b.keys[src];
        a.vals_in = b.vals[src];
        a.keys_out = b.keys[dst];
        a.vals_out = b.vals[dst];
        if (pass == 0)
            hipLaunchKernelGGL((onesweep_kernel<true, false>), dim3(tiles), dim3(256), 0, stream, a);
        else if (pass == 3)
            hipLaunchKernelGGL((onesweep_kernel<false, true>), dim3(tiles), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL((onesweep_kernel<false, false>), dim3(tiles), dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_sort_small_batch(const SortBatch& batch, uint32_t views, uint32_t capacity, hipStream_t stream)
{
    if (capacity == 0 || views == 0)
        return hipSuccess;
    if (capacity > kSmallSort)
        return hipErrorInvalidValue;
    const uint32_t lds = ((capacity + 3u) & ~3u) * 4;
    static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_small_batch_kernel),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
    if (raised != hipSuccess)
        return raised;
    hipLaunchKernelGGL(sort_small_batch_kernel, dim3((capacity + 63) / 64, views), dim3(256), lds, stream, batch);
    return hipGetLastError();
}

}  // namespace gv
