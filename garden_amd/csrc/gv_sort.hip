// gv_sort.hip — gv_sort: sortMeshes (source/system/render/mesh.cpp:265-328) on the device.
#include "gv_device.hpp"

namespace gv {

// ------------------------------------------------------------------------------------------------
// sortMeshes (mesh.cpp:265-328): order the compact records by distanceSq — ascending for unsorted buffers
// (front to back, operator< at render/mesh.hpp:196), descending for the sorted / translucent ones (:204).
// Stable LSD radix sort on a 32-bit order-preserving key, 8 bits per pass; ties keep ascending slot order
// (std::sort in the reference is unstable, so any tie order is within its contract).
// ------------------------------------------------------------------------------------------------
// Large pools: two launches per digit, nothing between them but the kernel boundary.
//   sort_rank_kernel     a workgroup ranks its tile of 4096 keys (stable; wave-private LDS counters), stores every key's
//                        rank among the tile's keys of the same digit (16 bits), the tile's 256 digit counts, and adds
//                        them into the counts of its GROUP of 32 tiles. Pass 0 builds the keys from distanceSq.
//   sort_scatter_kernel  reloads the tile's keys and ranks, gets the number of same-digit keys in all tiles before it
//                        as (group counts before its group) + (tile counts before it inside the group), and the digit's
//                        total as the sum over all groups — a few dozen INDEPENDENT loads, one or two memory round trips,
//                        no separate global histogram — reorders the tile by digit in LDS and writes
//                        each digit's run to its final place. The LAST pass does not write (key, index) pairs but gathers
//                        the 56-byte records straight to their sorted positions.
// Why not one launch per digit with a decoupled look-back (the round-2 first form, 223 us at 2.1 M records): with every
// tile resident at once the look-back is a serial chain through memory — 9-16 us of a 38 us pass were spent waiting for
// predecessors (tools/onesweep_probe.hip), a wider window only made it worse (64 words per step: 15.6 us median) — while
// a kernel boundary costs 1.6 us on this part and the keys, ranks and counts it has to carry (10 B per key) stay in L2 /
// Infinity Cache. No workgroup ever waits for another: static tile ids, no tickets, no polling.
#ifdef GV_SORT_TRACE  // dev tool only (tools/onesweep_probe.hip): wall-clock stamps per tile and phase
__device__ unsigned long long gv_sort_trace[4][8192][12];
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
constexpr uint32_t kSortTile = kSortTileKeys;      // keys per workgroup per pass ...
constexpr uint32_t kShortTile = kSortShortTileKeys;  // ... and for lists of up to kSortShortRecords records: the kernels are
                                                     // latency chains per tile whatever the count (rank 8 us, the record
                                                     // gather 26 us with six 4096-key tiles at 21 k records), so a short list
                                                     // is spread over four times as many workgroups with a quarter of the
                                                     // rounds each; long lists keep the long runs per digit (64-byte stores).
                                                     // Measured (tools/onesweep_probe.hip, short / long tiles): 21.7 k records
                                                     // 42 / 71 us, 65 k 46 / 78, 131 k 53 / 94, 308 k 64 / 100, 1 M 114 / 114,
                                                     // 2.1 M 234 / 204
constexpr uint32_t kSortGroup = kSortGroupTiles;   // tiles whose digit counts are also summed per group
#ifndef GV_SORT_CARRY_SLOTS  // 1: the pool slots travel with the (key, index) pairs
#define GV_SORT_CARRY_SLOTS 1
#endif
#ifndef GV_SORT_THREADS  // (tools/onesweep_probe.hip measures other workgroup sizes)
#define GV_SORT_THREADS 512
#endif
constexpr uint32_t kSortThreads = GV_SORT_THREADS;  // per workgroup of the rank / scatter kernels: the same tiles (run lengths) with
                                                    // half the rounds per lane and twice the waves per tile of the 256-thread form.
                                                    // Measured, 256 / 512 / 1024 threads: 1 M records 111 / 98 / 101 us, 2.1 M
                                                    // 198 / 190 / 207, 9.9 M 851 / 816 / 910
constexpr uint32_t kSortWaves = kSortThreads / 64;

__device__ __forceinline__ uint32_t order_key(float d, uint32_t descending)
{
    const uint32_t u = __float_as_uint(d);
    const uint32_t k = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);  // ascending float order == ascending key order
    return descending ? ~k : k;
}

// lanes of the wave whose `digit` equals this lane's (among `valid` lanes): 8 ballots
__device__ __forceinline__ unsigned long long match_digit(uint32_t digit, bool valid)
{
    // per bit: t = 0 / ~0 from the lane's bit, m = the lanes whose bit is set; the lanes that agree with this one are
    // ~(m ^ t) — six VALU operations per bit (sign-extending bit extract, compare, two xnor, two and) where the select form
    // of the same thing compiled to nine
    const unsigned long long all = __ballot(valid);
    uint32_t lo = (uint32_t)all, hi = (uint32_t)(all >> 32);
#pragma unroll
    for (uint32_t b = 0; b < 8; b++) {
        const uint32_t t = (uint32_t)((int32_t)(digit << (31u - b)) >> 31);
        const unsigned long long m = __ballot(t != 0u);
        lo &= ~((uint32_t)m ^ t);
        hi &= ~((uint32_t)(m >> 32) ^ t);
    }
    return ((unsigned long long)hi << 32) | lo;
}

// exclusive scan of one value per thread over the workgroup; every thread also gets the grand total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_sum /* LDS [kSortWaves] */, uint32_t* total)
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
    for (uint32_t w = 0; w < kSortWaves; w++) {
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
    uint32_t* group_hist;  // this sort's per-group digit counts [4][groups][256]   (zero on entry)
    uint32_t* next_set;    // the other parity's set, zeroed here for the next sort
    uint32_t set_words;
    uint32_t* tile_hist;   // [tiles][256]: digit counts of every tile, rewritten by each pass
    uint32_t groups;
};

struct SortPassArgs {
    const uint32_t* count;
    uint32_t capacity, descending, pass;
    uint32_t min_records;       // the radix kernels leave at once when the live count is not above this (the rank sort took it)
    const float* dist_in;       // FIRST: keys are built from these
    const uint32_t* keys_in;    // !FIRST
    const uint32_t* vals_in;    // !FIRST
    uint16_t* ranks;            // per key of the current order: rank among its tile's keys of the same digit
    uint32_t* keys_out;         // !LAST
    uint32_t* vals_out;         // !LAST
    const uint32_t* slots_in;   // !FIRST: the records' pool slots, in the current order (FIRST reads idx_in)
    uint32_t* slots_out;        // !LAST
    const uint32_t* idx_in;     // LAST: the records, gathered to their sorted positions
    const float* model_in;
    uint32_t* idx_out;
    float* model_out;
    float* dist_out;
    SortState st;
};

template <bool FIRST, uint32_t TILE>
__device__ __forceinline__ void sort_rank_tile(const SortPassArgs& a, const uint32_t n, uint32_t (*wcount)[256])
{
    constexpr uint32_t kSortTile = TILE, kSortRounds = TILE / kSortThreads;  // keys per workgroup, per lane
    const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
    if (blockIdx.x >= tiles)
        return;  // the grid is sized for the capacity, the count lives on the device: surplus workgroups leave at once
    const uint32_t tile = blockIdx.x;
    GV_TRACE(0)
    for (uint32_t k = threadIdx.x; k < kSortWaves * 256u; k += kSortThreads)
        (&wcount[0][0])[k] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t shift = a.pass * 8;
    // each wave ranks its own run of consecutive keys, 64 per round: wave-private counters, no workgroup barrier
    const uint32_t wave_base = tile * kSortTile + wave * (kSortTile / kSortWaves);
    uint32_t key[kSortRounds], local[kSortRounds];
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {  // all loads first
        const uint32_t j = wave_base + r * 64 + lane;
        const bool valid = j < n;
        if (FIRST)
            key[r] = valid ? order_key(a.dist_in[j], a.descending) : 0xFFFFFFFFu;
        else
            key[r] = valid ? a.keys_in[j] : 0xFFFFFFFFu;
    }
    // Ranking: per round, the lanes sharing my digit (a wave whose live keys all agree — the high bytes of one frame's
    // distances — knows without asking; otherwise 8 ballots), my rank among them, and the wave's running count of that
    // digit from a wave-private LDS table (the LDS executes a wave's instructions in order: round r sees exactly the
    // counts of rounds < r, so the ranking is stable). Two other forms were built and measured no faster on the box
    // (profiles/r02_sort_probe.txt): LDS lane-mask tables instead of the ballots (fewer instructions, more LDS round
    // trips), and all sixteen rounds' counts taken with returning LDS atomics in flight together (needs > 128 VGPRs).
    GV_TRACE_AFTER_LOADS(1)
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
    GV_TRACE(2)
    if (threadIdx.x < 256) {  // one thread per digit: the waves' counts become the counts of the waves before them, and
        const uint32_t d = threadIdx.x;  // their sum the tile's count, for the scatter kernel and for the tile's group
        uint32_t tile_count = 0;
#pragma unroll
        for (uint32_t w = 0; w < kSortWaves; w++) {
            const uint32_t c = wcount[w][d];
            wcount[w][d] = tile_count;
            tile_count += c;
        }
        a.st.tile_hist[(size_t)tile * 256 + d] = tile_count;
        if (tile_count)
            atomicAdd(&a.st.group_hist[((size_t)a.pass * a.st.groups + tile / kSortGroup) * 256 + d], tile_count);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {  // rank among the TILE's keys of the digit: the waves before mine come first
        const uint32_t j = wave_base + r * 64 + lane;
        if (j < n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            a.ranks[j] = (uint16_t)(local[r] + wcount[wave][d]);
        }
    }
    GV_TRACE(3)
}

template <bool FIRST>
__device__ __forceinline__ void sort_rank_body(const SortPassArgs& a)
{
    __shared__ uint32_t wcount[kSortWaves][256];  // per-wave digit counts of the tile
    const uint32_t n = min(*a.count, a.capacity);
    if (FIRST)  // the other parity's counters, for the next sort (nobody reads them during this one)
        for (uint32_t k = blockIdx.x * kSortThreads + threadIdx.x; k < a.st.set_words; k += gridDim.x * kSortThreads)
            a.st.next_set[k] = 0;
    if (n <= a.min_records)
        return;
    if (n <= kSortShortRecords)
        sort_rank_tile<FIRST, kShortTile>(a, n, wcount);
    else
        sort_rank_tile<FIRST, kSortTile>(a, n, wcount);
}

template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void sort_rank_kernel(const SortPassArgs a)
{
    sort_rank_body<FIRST>(a);
}

// Several lists per launch (launch_sort_batch): blockIdx.y picks the list, everything else is the single-list kernel — every list
// has its own buffers, counters and device-side count; a list shorter than the grid's widest leaves its surplus workgroups at once.
// N: entries in the argument (a launch pays for the size of its arguments).
template <uint32_t N>
struct SortPassBatch {
    SortPassArgs list[N];
};
template <bool FIRST, uint32_t N>
__global__ __launch_bounds__(kSortThreads) void sort_rank_batch_kernel(const SortPassBatch<N> batch)
{
    sort_rank_body<FIRST>(batch.list[blockIdx.y]);
}

struct SortScatterLds {
    uint32_t tile_excl[256];   // exclusive scan of the tile's digit counts (tile-local sorted order)
    uint32_t dst_base[256];    // global position of the tile's first key of each digit
    uint32_t skey[kSortTile];  // the tile reordered by digit
    uint32_t sval[kSortTile];
#if GV_SORT_CARRY_SLOTS
    uint32_t sslot[kSortTile];
#endif
    uint32_t wave_sum[kSortWaves];
};

template <bool FIRST, bool LAST, uint32_t TILE>
__device__ __forceinline__ void sort_scatter_tile(const SortPassArgs& a, const uint32_t n, SortScatterLds& lds)
{
    constexpr uint32_t kSortTile = TILE, kSortRounds = TILE / kSortThreads;
    uint32_t* const tile_excl = lds.tile_excl;
    uint32_t* const dst_base = lds.dst_base;
    uint32_t* const skey = lds.skey;
    uint32_t* const sval = lds.sval;
#if GV_SORT_CARRY_SLOTS
    uint32_t* const sslot = lds.sslot;
#endif
    uint32_t* const wave_sum = lds.wave_sum;
    const uint32_t tiles = (n + kSortTile - 1) / kSortTile;
    if (blockIdx.x >= tiles)
        return;
    const uint32_t tile = blockIdx.x;
    GV_TRACE(4)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t shift = a.pass * 8;
    const uint32_t wave_base = tile * kSortTile + wave * (kSortTile / kSortWaves);
    uint32_t key[kSortRounds], val[kSortRounds], rank[kSortRounds];
#if GV_SORT_CARRY_SLOTS
    uint32_t slot[kSortRounds];
#endif
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t j = wave_base + r * 64 + lane;
        const bool valid = j < n;
        if (FIRST) {
            key[r] = valid ? order_key(a.dist_in[j], a.descending) : 0xFFFFFFFFu;
            val[r] = j;
        } else {
            key[r] = valid ? a.keys_in[j] : 0xFFFFFFFFu;
            val[r] = valid ? a.vals_in[j] : 0u;
        }
#if GV_SORT_CARRY_SLOTS
        slot[r] = valid ? (FIRST ? (a.idx_in ? a.idx_in[j] : j) : a.slots_in[j]) : 0u;  // pass 0: a sequential read, where the last pass had a random one (no idx_in: the identity)
#endif
        rank[r] = valid ? (uint32_t)a.ranks[j] : 0u;
    }
    // one thread per digit: same-digit keys in the tiles before this one = whole groups + the tiles before it in its group
    const uint32_t d = threadIdx.x & 255u;
    const bool digit_thread = threadIdx.x < 256u;  // (workgroup-wave-uniform: whole waves)
    uint32_t tile_count = 0, before = 0, digit_total = 0;  // ... and the digit's count over ALL tiles (the global histogram, summed on the spot)
    if (digit_thread) {
        tile_count = a.st.tile_hist[(size_t)tile * 256 + d];
        const uint32_t group = tile / kSortGroup;
        const uint32_t* __restrict__ gh = a.st.group_hist + (size_t)a.pass * a.st.groups * 256 + d;
        const uint32_t* __restrict__ th = a.st.tile_hist + d;
        const uint32_t live_groups = (tiles + kSortGroup - 1) / kSortGroup;
#pragma unroll 8
        for (uint32_t g = 0; g < live_groups; g++) {
            const uint32_t c = gh[(size_t)g * 256];
            digit_total += c;
            before += g < group ? c : 0u;
        }
#pragma unroll 8
        for (uint32_t u = group * kSortGroup; u < tile; u++)
            before += th[(size_t)u * 256];
    }
    GV_TRACE_AFTER_LOADS(5)
    uint32_t total;
    const uint32_t gbase = block_exclusive_scan(digit_total, wave_sum, &total);  // keys of lower digits, all tiles
    const uint32_t texcl = block_exclusive_scan(tile_count, wave_sum, &total);                   // ... in this tile
    if (digit_thread) {
        tile_excl[d] = texcl;
        dst_base[d] = gbase + before;
    }
    __syncthreads();
    GV_TRACE(6)
    // reorder the tile by digit in LDS, then write every digit's run to its place
#pragma unroll
    for (uint32_t r = 0; r < kSortRounds; r++) {
        const uint32_t j = wave_base + r * 64 + lane;
        if (j < n) {
            const uint32_t lpos = tile_excl[(key[r] >> shift) & 255u] + rank[r];
            skey[lpos] = key[r];
            sval[lpos] = val[r];
#if GV_SORT_CARRY_SLOTS
            sslot[lpos] = slot[r];
#endif
        }
    }
    __syncthreads();
    GV_TRACE(7)
    const uint32_t live = min(kSortTile, n - tile * kSortTile);
    if (!LAST) {
#pragma unroll 4
        for (uint32_t t = threadIdx.x; t < live; t += kSortThreads) {
            const uint32_t k = skey[t], v = sval[t];
            const uint32_t dd = (k >> shift) & 255u;
            const uint32_t pos = dst_base[dd] + (t - tile_excl[dd]);
            a.keys_out[pos] = k;
            a.vals_out[pos] = v;
#if GV_SORT_CARRY_SLOTS
            a.slots_out[pos] = sslot[t];
#endif
        }
        GV_TRACE(8)
        return;
    }
    // LAST: the records go straight to their sorted positions. distanceSq is the key itself (order_key is a bijection:
    // no gather), the pool slot has travelled with the pair (as a 4-byte gather from idx_in it cost 27 us of a 97 us pass at
    // 2.1 M records: a whole sector fetched per slot; carried, it costs the other passes ~1.5 us each with 512-lane
    // workgroups — with 256-lane ones it cost them 4-7 us each and was not worth it), the 48-byte model is gathered by three
    // lanes per record (one float4 each), so that the stores of a run of records are whole contiguous rows.
    for (uint32_t t = threadIdx.x; t < live; t += kSortThreads) {
        const uint32_t k = skey[t];
        const uint32_t dd = (k >> shift) & 255u;
        const uint32_t pos = dst_base[dd] + (t - tile_excl[dd]);
        const uint32_t u = a.descending ? ~k : k;
        if (a.dist_out)
            a.dist_out[pos] = __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
#if GV_SORT_CARRY_SLOTS
        a.idx_out[pos] = sslot[t];
#else
        a.idx_out[pos] = a.idx_in ? a.idx_in[sval[t]] : sval[t];  // sval[t] = the record's index before the sort
#endif
        skey[t] = pos;  // only this thread reads skey[t] in this loop
    }
    if (!a.model_in)  // (uniform) a sort of bare keys: the order is the result (launch_sort_keys)
        return;
    __syncthreads();
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a.model_in);
    float4* __restrict__ dst = reinterpret_cast<float4*>(a.model_out);
#pragma unroll 4
    for (uint32_t q = threadIdx.x; q < live * 3u; q += kSortThreads) {
        const uint32_t t = q / 3u, part = q - t * 3u;
        dst[(size_t)skey[t] * 3 + part] = src[(size_t)sval[t] * 3 + part];
    }
    GV_TRACE(8)
}

template <bool FIRST, bool LAST>
__device__ __forceinline__ void sort_scatter_body(const SortPassArgs& a)
{
    __shared__ SortScatterLds lds;
    const uint32_t n = min(*a.count, a.capacity);
    if (n <= a.min_records)
        return;
    if (n <= kSortShortRecords)
        sort_scatter_tile<FIRST, LAST, kShortTile>(a, n, lds);
    else
        sort_scatter_tile<FIRST, LAST, kSortTile>(a, n, lds);
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const SortPassArgs a)
{
    sort_scatter_body<FIRST, LAST>(a);
}
template <bool FIRST, bool LAST, uint32_t N>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_batch_kernel(const SortPassBatch<N> batch)
{
    sort_scatter_body<FIRST, LAST>(batch.list[blockIdx.y]);
}

// Small pools (up to kSmallSort records possible): ONE launch instead of fourteen — a tick of an engine-sized scene
// (10^4 entities) is launch-bound. Rank sort: the position of record i is the number of records that order before
// it, (key, emission index) compared as a pair, so the result is the stable order the radix passes produce. A
// workgroup owns 64 records; its four waves each count over a quarter of the keys (staged in LDS, read as uniform
// 16-byte broadcasts) and the partial counts meet in LDS. n^2 / 4 comparisons per wave, all CUs busy, no
// inter-workgroup step: ~6 us at 2 k records where a one-workgroup bitonic network took 42 us.
constexpr uint32_t kSmallSort = kBatchSortMaxSlots;
constexpr uint32_t kMidSortSlots = 1u << 20;  // pools up to this size also get the rank-sort launch (see launch_sort)

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

// What publish_kernel (gv_cull.hip) would do for this view after the sort, done by the sort's own launch: the count and (main
// pass) the isVisible bytes in pool-slot order go to the host here, shared by all workgroups of the view's row of the grid
// — the records follow from sort_small_block, each at its sorted place. One kernel boundary and the publish kernel's
// dependent loads (count, records) less per tick.
__device__ __forceinline__ void publish_visibility(const PublishArgs& a, uint32_t n, uint32_t block, uint32_t nblocks)
{
    if (block == 0 && threadIdx.x == 0)
        *a.host_count = n;
    if (!a.host_is_visible)
        return;
    if (a.orig) {  // spatially ordered mirror: back into pool-slot order in LDS, out as whole words — by the row's LAST
        if (block != nblocks - 1)  // workgroup: it has no records to sort unless nearly every slot is visible, so the
            return;                // un-permutation runs beside the sort instead of in front of workgroup 0's share of it
        __shared__ uint8_t slots[(kBatchSortMaxSlots + 3u) & ~3u];
        for (uint32_t j = threadIdx.x; j < a.occupancy; j += 256)
            slots[a.orig[j]] = a.is_visible[j];
        __syncthreads();
        const uint32_t words = a.occupancy >> 2;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(slots);
        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(a.host_is_visible);
        for (uint32_t w = threadIdx.x; w < words; w += 256)
            dst[w] = src[w];
        for (uint32_t j = (words << 2) + threadIdx.x; j < a.occupancy; j += 256)
            a.host_is_visible[j] = slots[j];
        __syncthreads();
    } else {
        const uint32_t tid = block * 256 + threadIdx.x, threads = nblocks * 256;
        const uint32_t words = a.occupancy >> 2;
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(a.is_visible);
        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(a.host_is_visible);
        for (uint32_t w = tid; w < words; w += threads)
            dst[w] = src[w];
        for (uint32_t j = (words << 2) + tid; j < a.occupancy; j += threads)
            a.host_is_visible[j] = a.is_visible[j];
    }
}

// One sorted record to the host, at its sorted place: the three arrays, or the caller's record struct (GvRecordLayout; built in
// the lane's own row of `stage` so that the bytes no field owns leave as zeros, written out as whole 16-byte pieces).
__device__ __forceinline__ void publish_record(const PublishArgs& a, uint32_t rank, uint32_t idx, float dist, float4 m0, float4 m1, float4 m2,
                                               uint32_t* stage /* LDS: kMaxRecordStride / 4 words of this lane */)
{
    if (a.host_records) {
        const RecordLayout& L = a.layout;
        const uint32_t words = L.stride >> 2;
        for (uint32_t k = 0; k < words; k++)
            stage[k] = 0;
        const unsigned long long offset = (unsigned long long)record_slot(L, idx) * L.component_stride;  // componentOffset  mesh.cpp:170
        stage[L.component_offset >> 2] = (uint32_t)offset;
        stage[(L.component_offset >> 2) + 1] = (uint32_t)(offset >> 32);
        uint32_t* bm = stage + (L.baked_model >> 2);
        bm[0] = __float_as_uint(m0.x); bm[1] = __float_as_uint(m0.y); bm[2] = __float_as_uint(m0.z); bm[3] = __float_as_uint(m0.w);
        bm[4] = __float_as_uint(m1.x); bm[5] = __float_as_uint(m1.y); bm[6] = __float_as_uint(m1.z); bm[7] = __float_as_uint(m1.w);
        bm[8] = __float_as_uint(m2.x); bm[9] = __float_as_uint(m2.y); bm[10] = __float_as_uint(m2.z); bm[11] = __float_as_uint(m2.w);
        stage[L.distance_sq >> 2] = __float_as_uint(dist);
        if (L.buffer_index != 0xFFFFFFFFu)
            stage[L.buffer_index >> 2] = L.buffer_index_value;
        uint4* out = reinterpret_cast<uint4*>(a.host_records + (size_t)rank * L.stride);
        for (uint32_t q = 0; q < (L.stride >> 4); q++)
            out[q] = make_uint4(stage[4 * q], stage[4 * q + 1], stage[4 * q + 2], stage[4 * q + 3]);
    } else if (a.host_idx) {
        a.host_idx[rank] = idx;
        a.host_dist[rank] = dist;
        float4* hm = reinterpret_cast<float4*>(a.host_model) + (size_t)rank * 3;
        hm[0] = m0;
        hm[1] = m1;
        hm[2] = m2;
    }
}

// The same count with the keys read from memory (built from distanceSq on the fly; uniform 16-byte loads): for a list that
// outgrew the LDS key table in a launch that has no radix passes behind it (kSortRankOnly) — correct at any count, several
// times slower than the table form, and rare: the count has to jump past the table within one frame.
template <int WHERE>
__device__ __forceinline__ uint32_t count_before_global(const float* __restrict__ dist, uint32_t descending, uint32_t n, uint32_t jlo, uint32_t jhi,
                                                        uint32_t ki, uint32_t i)
{
    uint32_t before = 0;
#pragma unroll 2
    for (uint32_t j = jlo; j < jhi; j += 4) {
        uint4 k;
        if (j + 3u < n) {
            const float4 d = *reinterpret_cast<const float4*>(dist + j);
            k = make_uint4(order_key(d.x, descending), order_key(d.y, descending), order_key(d.z, descending), order_key(d.w, descending));
        } else {  // the last, partly filled group: padding keys are never counted
            k.x = j < n ? order_key(dist[j], descending) : 0xFFFFFFFFu;
            k.y = j + 1u < n ? order_key(dist[j + 1u], descending) : 0xFFFFFFFFu;
            k.z = j + 2u < n ? order_key(dist[j + 2u], descending) : 0xFFFFFFFFu;
            k.w = 0xFFFFFFFFu;
        }
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

// max_records: a live count above it leaves the records to the radix kernels behind; table: the key table's capacity (LDS) —
// counts between the two take the from-memory form above
template <class Entry>
__device__ __forceinline__ void sort_small_block(const Entry& b, uint32_t capacity, uint32_t descending, uint32_t block,
                                                 uint32_t max_records = 0xFFFFFFFFu, const PublishArgs* pub = nullptr, uint32_t table = 0xFFFFFFFFu)
{
    extern __shared__ uint32_t key[];  // order-preserving keys of all n records, padded to a multiple of 4
    __shared__ uint32_t partial[4][64];
    // the first 2048 distances are asked for together with the count they depend on (the buffers are sized for the pool): a
    // short list's keys cost one round trip to memory instead of two
    constexpr uint32_t kSpeculated = 8;
    float early[kSpeculated];
#pragma unroll
    for (uint32_t k = 0; k < kSpeculated; k++)
        early[k] = b.dist_in[min(threadIdx.x + 256u * k, capacity - 1u)];
    const uint32_t n = min(*b.count, capacity);
    if (pub)  // (uniform)
        publish_visibility(*pub, n, block, gridDim.x);
    const uint32_t i0 = block * 64;
    if (i0 >= n || n > max_records)
        return;
    const uint32_t n4 = (n + 3u) & ~3u;
    const bool in_lds = n <= table;  // (uniform)
    if (in_lds) {
#pragma unroll
        for (uint32_t r = 0; r < kSpeculated; r++) {
            const uint32_t j = threadIdx.x + 256u * r;
            if (j < n4)
                key[j] = j < n ? order_key(early[r], descending) : 0xFFFFFFFFu;  // padding: never counted (its index is beyond every record's)
        }
        for (uint32_t j = threadIdx.x + 256u * kSpeculated; j < n4; j += 256)
            key[j] = j < n ? order_key(b.dist_in[j], descending) : 0xFFFFFFFFu;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const uint32_t i = i0 + lane;
    const uint32_t per = ((n4 >> 2) + 3u) & ~3u;  // keys per wave, a multiple of 4
    const uint32_t jlo = min(part * per, n4), jhi = min(jlo + per, n4);
    const uint32_t own_lo = min(max(i0 & ~3u, jlo), jhi), own_hi = min(max((i0 + 64u + 3u) & ~3u, jlo), jhi);
    uint32_t before;
    if (in_lds) {
        const uint32_t ki = key[min(i, n4 - 1)];
        before = count_before<0>(key, jlo, own_lo, ki, i) + count_before<1>(key, own_lo, own_hi, ki, i) + count_before<2>(key, own_hi, jhi, ki, i);
    } else {
        const uint32_t ki = order_key(b.dist_in[min(i, n - 1u)], descending);
        before = count_before_global<0>(b.dist_in, descending, n, jlo, own_lo, ki, i) +
                 count_before_global<1>(b.dist_in, descending, n, own_lo, own_hi, ki, i) +
                 count_before_global<2>(b.dist_in, descending, n, own_hi, jhi, ki, i);
    }
    partial[part][lane] = before;
    __syncthreads();
    if (part != 0 || i >= n)
        return;
    const uint32_t rank = partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
    const uint32_t idx = b.idx_in ? b.idx_in[i] : i;
    const float dist = b.dist_in[i];
    b.idx_out[rank] = idx;
    if (b.dist_out)
        b.dist_out[rank] = dist;
    if (!b.model_in)  // (uniform) bare keys
        return;
    const float4* sm = reinterpret_cast<const float4*>(b.model_in + (size_t)i * 12);
    float4* dm = reinterpret_cast<float4*>(b.model_out + (size_t)rank * 12);
    const float4 m0 = sm[0], m1 = sm[1], m2 = sm[2];
    dm[0] = m0;
    dm[1] = m1;
    dm[2] = m2;
    if (pub) {
        __shared__ uint32_t record_stage[64][kMaxRecordStride / 4 + 1];  // (+1: rows on different banks)
        publish_record(*pub, rank, idx, dist, m0, m1, m2, record_stage[lane]);
    }
}

__global__ __launch_bounds__(256) void sort_small_kernel(const SortBuffers b, uint32_t capacity, uint32_t descending, uint32_t max_records,
                                                         uint32_t table)
{
    sort_small_block(b, capacity, descending, blockIdx.x, max_records, nullptr, table);
}

// several views of one small pool (main camera + shadow passes) in one launch: blockIdx.y picks the view. N: entries in the
// argument — a launch pays for the size of its arguments (2.4 us with a small one, 5.8 us at 20 KB), and the 32-entry batch is
// 6.4 KB where a tick of one mesh system uses 200 bytes of it
template <uint32_t N>
struct SortBatchN {
    SmallSortEntry view[N];
};
template <uint32_t N>
__global__ __launch_bounds__(256) void sort_small_batch_kernel(const SortBatchN<N> batch)
{
    const SmallSortEntry& e = batch.view[blockIdx.y];
    sort_small_block(e, e.capacity, e.descending, blockIdx.x, 0xFFFFFFFFu, e.fused_publish ? &e.publish : nullptr);
}

// the rank-sort launch of launch_sort for several mid-sized lists at once (launch_sort_batch): blockIdx.y picks the list
struct MidRankEntry {
    SortBuffers b;
    uint32_t capacity, descending, max_records, table;
};
template <uint32_t N>
struct MidRankBatch {
    MidRankEntry list[N];
};
template <uint32_t N>
__global__ __launch_bounds__(256) void sort_mid_rank_batch_kernel(const MidRankBatch<N> batch)
{
    const MidRankEntry& e = batch.list[blockIdx.y];
    sort_small_block(e.b, e.capacity, e.descending, blockIdx.x, e.max_records, nullptr, e.table);
}

// what launch_sort decides on the host for one list: whether it gets the rank-sort launch (and with which key table), whether the
// radix launches, and what the device-side count leaves to which of the two
struct SortPlan {
    bool rank_only, rank_sort;
    uint32_t rank_records;  // the rank-sort launch's key table (LDS words)
    uint32_t rank_blocks;   // its workgroups
};
static SortPlan sort_plan(uint32_t capacity, SortMode mode)
{
    SortPlan p{};
    p.rank_only = sort_is_rank_only(capacity, mode);
    p.rank_sort = capacity <= kMidSortSlots && (mode != kSortRadixOnly || capacity <= kSmallSort);
    p.rank_records = capacity <= kSmallSort ? capacity : (p.rank_only ? kRankOnlyTableRecords : kRankSortMaxRecords);
    p.rank_blocks = ((p.rank_only ? capacity : p.rank_records) + 63) / 64;
    return p;
}

static void sort_pass_args(const SortBuffers& b, uint32_t capacity, bool descending, bool rank_sort, SortPassArgs& a)
{
    a = SortPassArgs{};
    a.st.groups = sort_group_count(capacity);
    a.st.set_words = sort_set_words(capacity);
    a.st.group_hist = b.counters[b.parity];
    a.st.next_set = b.counters[b.parity ^ 1u];
    a.st.tile_hist = b.tile_hist;
    a.count = b.count;
    a.capacity = capacity;
    a.descending = descending ? 1u : 0u;
    a.min_records = rank_sort ? kRankSortMaxRecords : 0u;
    a.dist_in = b.dist_in;
    a.idx_in = b.idx_in;
    a.model_in = b.model_in;
    a.idx_out = b.idx_out;
    a.model_out = b.model_out;
    a.dist_out = b.dist_out;
    a.ranks = b.ranks;
}
static void sort_pass_buffers(const SortBuffers& b, uint32_t pass, SortPassArgs& a)
{
    const uint32_t src = (pass & 1u) ^ 1u, dst = pass & 1u;  // pass 0 writes set 0, pass 1 set 1, ...
    a.pass = pass;
    a.keys_in = b.keys[src];
    a.vals_in = b.vals[src];
    a.keys_out = b.keys[dst];
    a.vals_out = b.vals[dst];
    a.slots_in = b.slots[src];
    a.slots_out = b.slots[dst];
}

template <uint32_t N>
static hipError_t launch_sort_batch_n(const SortBatchEntry* lists, uint32_t count, hipStream_t stream)
{
    // the rank-sort launch for every list that gets one (a list that does not: max_records 0 — its workgroups leave after one load)
    MidRankBatch<N> ranks{};
    uint32_t rank_blocks = 0, rank_lds = 0, radix_tiles = 0;
    SortPassBatch<N> passes{};
    for (uint32_t k = 0; k < count; k++) {
        const SortBatchEntry& e = lists[k];
        const SortPlan p = sort_plan(e.capacity, e.mode);
        ranks.list[k] = MidRankEntry{e.b, e.capacity, e.descending, p.rank_sort ? (p.rank_only ? 0xFFFFFFFFu : p.rank_records) : 0u, p.rank_records};
        if (p.rank_sort) {
            rank_blocks = std::max(rank_blocks, p.rank_blocks);
            rank_lds = std::max(rank_lds, ((p.rank_records + 3u) & ~3u) * 4u);
        }
        sort_pass_args(e.b, e.capacity, e.descending != 0, p.rank_sort, passes.list[k]);
        if (p.rank_only)
            passes.list[k].min_records = 0xFFFFFFFFu;  // the rank-sort launch alone: the radix launches leave this list at once
        else
            radix_tiles = std::max(radix_tiles, sort_tile_count(e.capacity));
    }
    if (rank_blocks) {
        static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_mid_rank_batch_kernel<N>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
        if (raised != hipSuccess)
            return raised;
        hipLaunchKernelGGL(sort_mid_rank_batch_kernel<N>, dim3(rank_blocks, count), dim3(256), rank_lds, stream, ranks);
    }
    if (!radix_tiles)
        return hipGetLastError();
    const dim3 grid(radix_tiles, count), block(kSortThreads);
    for (uint32_t pass = 0; pass < 4; pass++) {
        for (uint32_t k = 0; k < count; k++)
            sort_pass_buffers(lists[k].b, pass, passes.list[k]);
        if (pass == 0) {
            hipLaunchKernelGGL((sort_rank_batch_kernel<true, N>), grid, block, 0, stream, passes);
            hipLaunchKernelGGL((sort_scatter_batch_kernel<true, false, N>), grid, block, 0, stream, passes);
        } else {
            hipLaunchKernelGGL((sort_rank_batch_kernel<false, N>), grid, block, 0, stream, passes);
            if (pass == 3)
                hipLaunchKernelGGL((sort_scatter_batch_kernel<false, true, N>), grid, block, 0, stream, passes);
            else
                hipLaunchKernelGGL((sort_scatter_batch_kernel<false, false, N>), grid, block, 0, stream, passes);
        }
    }
    return hipGetLastError();
}

hipError_t launch_sort_batch(const SortBatchEntry* lists, uint32_t count, hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    if (count == 1)
        return launch_sort(lists[0].b, lists[0].capacity, lists[0].descending != 0, stream, lists[0].mode);
    return count <= 8 ? launch_sort_batch_n<8>(lists, count, stream) : launch_sort_batch_n<kMaxSortBatch>(lists, count, stream);
}

hipError_t launch_sort(const SortBuffers& b, uint32_t capacity, bool descending, hipStream_t stream, SortMode mode)
{
    if (capacity == 0)
        return hipSuccess;
    // kSortRankOnly (the caller expects a short list: the previous frame's count): a mid-sized pool gets the rank-sort launch
    // alone — the eight radix launches that would leave after one load each are ~16 us of an engine-sized tick — and a
    // list that outgrew the key table after all is still sorted by it, from memory (count_before_global).
    // Short lists sort in one launch whatever the pool's size: a pool of up to kMidSortSlots slots (where only the device
    // knows how short the visible list is) gets the rank-sort launch AND the radix launches, and the live count decides on
    // the device which of the two does the work — the other leaves after one load, ~2 us per launch, against 70 us for
    // the eight radix launches on a few thousand records. Measured crossover: ~12 k records (rank 11 us at 2 k, 75 us at
    // 16 k, 160 us at 32 k).
    const SortPlan plan = sort_plan(capacity, mode);
    if (plan.rank_sort) {
        const uint32_t lds = ((plan.rank_records + 3u) & ~3u) * 4;  // the key table, beside the 1 KB of partial counts
        static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_small_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
        if (raised != hipSuccess)
            return raised;
        hipLaunchKernelGGL(sort_small_kernel, dim3(plan.rank_blocks), dim3(256), lds, stream, b, capacity, descending ? 1u : 0u,
                           plan.rank_only ? 0xFFFFFFFFu : plan.rank_records, plan.rank_records);
        if (capacity <= kSmallSort || plan.rank_only)
            return hipGetLastError();
    }
    const uint32_t tiles = sort_tile_count(capacity);  // at full capacity (long or short tiles); the live count is on the device
    SortPassArgs a;
    sort_pass_args(b, capacity, descending, plan.rank_sort, a);
    for (uint32_t pass = 0; pass < 4; pass++) {
        sort_pass_buffers(b, pass, a);
        if (pass == 0) {
            hipLaunchKernelGGL(sort_rank_kernel<true>, dim3(tiles), dim3(kSortThreads), 0, stream, a);
            hipLaunchKernelGGL((sort_scatter_kernel<true, false>), dim3(tiles), dim3(kSortThreads), 0, stream, a);
        } else {
            hipLaunchKernelGGL(sort_rank_kernel<false>, dim3(tiles), dim3(kSortThreads), 0, stream, a);
            if (pass == 3)
                hipLaunchKernelGGL((sort_scatter_kernel<false, true>), dim3(tiles), dim3(kSortThreads), 0, stream, a);
            else
                hipLaunchKernelGGL((sort_scatter_kernel<false, false>), dim3(tiles), dim3(kSortThreads), 0, stream, a);
        }
    }
    return hipGetLastError();
}

__global__ void done_flag_kernel(uint32_t* host_flag, uint32_t value)
{
    __hip_atomic_store(host_flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_done_flag(uint32_t* host_flag, uint32_t value, hipStream_t stream)
{
    hipLaunchKernelGGL(done_flag_kernel, dim3(1), dim3(1), 0, stream, host_flag, value);
    return hipGetLastError();
}

hipError_t launch_sort_small_batch(const SortBatch& batch, uint32_t views, uint32_t capacity, hipStream_t stream)
{
    if (capacity == 0 || views == 0)
        return hipSuccess;
    if (capacity > kSmallSort)
        return hipErrorInvalidValue;
    const uint32_t lds = ((capacity + 3u) & ~3u) * 4;
    static_assert(sizeof(SortBatchN<kMaxSortViews>) == sizeof(SortBatch), "the full batch is the 32-entry form");
    static const hipError_t raised4 = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_small_batch_kernel<4>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
    static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void*>(sort_small_batch_kernel<kMaxSortViews>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSort * 4);
    if (raised != hipSuccess || raised4 != hipSuccess)
        return raised != hipSuccess ? raised : raised4;
    const dim3 grid((capacity + 63) / 64, views);
    if (views <= 4) {
        SortBatchN<4> few;
        for (uint32_t k = 0; k < 4; k++)
            few.view[k] = batch.view[k];
        hipLaunchKernelGGL(sort_small_batch_kernel<4>, grid, dim3(256), lds, stream, few);
    } else {
        SortBatchN<kMaxSortViews> all;
        for (uint32_t k = 0; k < kMaxSortViews; k++)
            all.view[k] = batch.view[k];
        hipLaunchKernelGGL(sort_small_batch_kernel<kMaxSortViews>, grid, dim3(256), lds, stream, all);
    }
    return hipGetLastError();
}

}  // namespace gv
