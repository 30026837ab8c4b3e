// sample.hip -- neighbour sampling of a GPU-resident CSR graph on gfx950.
// Same contract as sampler.sample_adj (cogdl/operators/sample/sample.cpp:6-144), which the reference runs
// single-threaded on the CPU (libc rand(), an unordered_map for the relabelling) inside DataLoader workers:
//   seeds get the local ids 0..batch-1, every other sampled node the next id in DISCOVERY order (seed rows in order,
//   a row's sampled edges in emission order); out_indices holds local ids, out_edges the CSR positions.
// Pipeline (all stream-ordered, no host round trip, no atomics => deterministic for a given seed):
//   1. counts per seed + exclusive scan                                  -> out_indptr
//   2. one WAVE per seed picks its edges and appends (node id, position) to a flat sequence Q = [seeds | picks]
//        all / with replacement: lanes work independently (counter-based RNG: hash(seed, row, draw))
//        without replacement:    Floyd's algorithm, the membership test of each draw is one wave-wide ballot over
//                                the chosen set in LDS; the chosen positions are emitted in ascending order
//   3. discovery-order relabelling: the local id of a node is the number of FIRST occurrences in Q before its own first
//      occurrence.  Round 3 (default): the pick kernel records every node's smallest position in Q in a hash table
//      (open addressing; slot claimed by an integer compare-and-swap, position by an integer atomic minimum -- the
//      table's CONTENT as a map node -> first position does not depend on the order of the insertions, so the outputs
//      stay deterministic); kernel `first` looks every position's node up, flags first occurrences and ranks them inside
//      its 16 k-position block; kernel `relabel` adds the block prefixes.  Four launches per call (three when Q fits one
//      block) instead of the ~25 of the sort-based form -- a captured mini-batch step is bound by its count of dependent
//      kernel nodes, not by their work.
//      Round 1-2 form (tuning key 11 = 1, kept for A/B tests and for frontiers beyond 6.7e7 positions): stable sort of the positions
//      in Q by node id (rocPRIM's radix sort until round 5, this library's radix transpose of a one-row matrix since);
//      the head of every run of equal ids is that node's first occurrence; an exclusive scan of the first-occurrence
//      flags over Q gives the local ids; a running maximum over the sorted order hands every duplicate its head.
// Integer work, latency bound at mini-batch sizes; int64 in and out like the reference.
#include "common.h"

#include "scan.h"

namespace cogdl {

// The sort of the round 1-2 form is this library's own since round 5 (transpose.hip: sort_positions_by_key = the radix
// transpose of a one-row matrix; kernels only, so the fixed-capacity entry point stays capturable in a hipGraph at any
// size -- rocPRIM's onesweep clears its state with hipMemsetAsync, memset nodes that were seen not to replay).
size_t sort_positions_workspace_bytes(int64_t n_keys, int64_t len);
int sort_positions_by_key(const uint32_t *keys, int64_t len, int64_t n_keys, int32_t *perm_out, void *workspace,
                          size_t workspace_bytes, hipStream_t s);

// skeys[j] = keys[spos[j]] (the sorted keys of the position sort)
__global__ void sample_gather_keys_kernel(const uint32_t *__restrict__ keys, const int32_t *__restrict__ spos,
                                          uint32_t *__restrict__ skeys, int64_t len) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += (int64_t)gridDim.x * blockDim.x)
        skeys[j] = keys[spos[j]];
}
constexpr int kSampleMaxK = 1024;  // without replacement: chosen set of one seed lives in LDS

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

static unsigned sample_key_bits(int64_t num_nodes) {  // one more than the ids need: the padding key 1 << bits
    unsigned b = 1;
    while (b < 32 && (int64_t(1) << b) < num_nodes) ++b;
    return b;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// draw `j` of seed row `i`: a pure function of (seed, i, j) -- independent of scheduling
__device__ __forceinline__ uint64_t draw(uint64_t seed, uint64_t i, uint64_t j) { return mix64(mix64(seed ^ mix64(i)) + j); }

__device__ __forceinline__ int64_t count_for(int64_t deg, int64_t k, int replace) {
    if (k < 0) return deg;
    if (replace) return deg > 0 ? k : 0;
    return deg < k ? deg : k;
}

// flags: bit 0 = a seed outside [0, num_nodes), bit 1 = a neighbour id outside, bit 2 = output capacity exceeded
// `batch_count` (device, may be NULL): how many of the `batch` seed slots are in use -- the padded entry point
// samples into buffers of a FIXED capacity, so that a whole mini-batch step can be captured in a hipGraph.
__device__ __forceinline__ int64_t valid_seeds(const int64_t *batch_count, int64_t batch) {
    if (!batch_count) return batch;
    const int64_t v = *batch_count;
    return v < 0 ? 0 : (v > batch ? batch : v);
}

// ---------------------------------------------------------------------------------------------------------------------
// Hash-table relabelling (see the file header, step 3).
constexpr uint32_t kHtEmpty = 0xFFFFFFFFu;     // no node id equals it (ids < 2^31)
constexpr uint32_t kHtNoPos = 0x7FFFFFFFu;
constexpr int kRlThreads = 1024;
constexpr int kRlItems1 = 16, kRlTile1 = kRlThreads * kRlItems1;  // positions of Q of the single-workgroup form
constexpr int kRlItems = 8, kRlTile = kRlThreads * kRlItems;     // ... per workgroup of the several-block form (register budget: 128)
constexpr int64_t kRlMaxBlocks = 8192;         // block prefixes are scanned in LDS by every workgroup

__device__ __forceinline__ uint32_t ht_hash(uint32_t k) {  // murmur3 finaliser
    k ^= k >> 16;
    k *= 0x85EBCA6Bu;
    k ^= k >> 13;
    k *= 0xC2B2AE35u;
    k ^= k >> 16;
    return k;
}
struct HashTable {
    uint32_t *key, *pos;
    uint32_t mask;  // capacity - 1 (capacity: a power of two >= 2 * |Q|)
};
// -> the slot that holds `key` (the position of Q that inserted it keeps the slot: the look-up later is ONE load of
// pos[slot] per position, all of a thread's loads in flight at once, instead of a dependent probe sequence each)
__device__ __forceinline__ uint32_t ht_insert(const HashTable &t, uint32_t key, uint32_t pos) {
    uint32_t h = ht_hash(key) & t.mask;
    for (;;) {
        const uint32_t cur = atomicCAS(&t.key[h], kHtEmpty, key);
        if (cur == kHtEmpty || cur == key) {
            atomicMin(&t.pos[h], pos);
            return h;
        }
        h = (h + 1) & t.mask;
    }
}

// Exclusive scan of one value per thread over a 1024-thread workgroup; returns the exclusive prefix, *total = the sum.
// `sh` = 17 words of LDS; two barriers.
template <typename V>
__device__ __forceinline__ V wg1024_exclusive_scan(V v, V *sh, V *total) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
    V incl = v;
#pragma unroll
    for (int sft = 1; sft < kWave; sft <<= 1) {
        const V u = __shfl_up(incl, sft, kWave);
        if (lane >= sft) incl += u;
    }
    __syncthreads();  // (a previous use of sh is over)
    if (lane == kWave - 1) sh[w] = incl;
    __syncthreads();
    V off = 0, tot = 0;
#pragma unroll
    for (int ww = 0; ww < kRlThreads / kWave; ++ww) {
        const V x = sh[ww];
        if (ww < w) off += x;
        tot += x;
    }
    *total = tot;
    return off + incl - v;
}

// N values per thread in the STRIDED layout (value q of thread t is element q * 1024 + t of a block of N * 1024: a wave's
// loads and stores are then contiguous 256-byte runs -- with N consecutive elements per thread every memory instruction
// of a wave touches 64 different lines and the address unit serialises it: 20 us per kernel on one CU, measured):
// v[q] is replaced by the exclusive prefix of its element in block order; *total = the block's sum.  `sh` = 16 * N + 5
// words of LDS; four barriers.
template <int N, typename V>
__device__ __forceinline__ void wg1024_strided_exclusive_scan(V (&v)[N], V *sh, V *total) {
    static_assert(16 * N <= 256, "the wave totals are scanned by the first four waves");
    const int t = threadIdx.x, lane = t & (kWave - 1), w = t >> 6;
    V incl[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        V x = v[q];
#pragma unroll
        for (int sft = 1; sft < kWave; sft <<= 1) {
            const V u = __shfl_up(x, sft, kWave);
            if (lane >= sft) x += u;
        }
        incl[q] = x;
    }
    __syncthreads();  // (a previous use of sh is over)
    if (lane == kWave - 1) {
#pragma unroll
        for (int q = 0; q < N; ++q) sh[q * 16 + w] = incl[q];
    }
    __syncthreads();
    // the 16 * N wave totals, in block order (q major), scanned by the first four waves
    V mine = 0, x = 0;
    if (t < 256) {
        mine = t < 16 * N ? sh[t] : V(0);
        x = mine;
#pragma unroll
        for (int sft = 1; sft < kWave; sft <<= 1) {
            const V u = __shfl_up(x, sft, kWave);
            if (lane >= sft) x += u;
        }
        if (lane == kWave - 1) sh[16 * N + w] = x;
    }
    __syncthreads();
    if (t < 256) {
        V off = 0;
        for (int ww = 0; ww < w; ++ww) off += sh[16 * N + ww];
        if (t < 16 * N) sh[t] = off + x - mine;  // (read into `mine` before the barrier above)
        if (t == 16 * N - 1) sh[16 * N + 4] = off + x;
    }
    __syncthreads();
    *total = sh[16 * N + 4];
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = sh[q * 16 + w] + incl[q] - v[q];
}

typedef int64_t I64x2 __attribute__((ext_vector_type(2), aligned(8)));

// prep: block 0 -- flags = 0, per-seed counts and their exclusive scan -> out_indptr (int64; out_indptr[batch] = total);
// every block -- its share of the hash table set to empty.
__global__ __launch_bounds__(kRlThreads) void sample_prep_kernel(const int64_t *__restrict__ indptr,
                                                                 const int64_t *__restrict__ node_idx, int64_t batch,
                                                                 const int64_t *__restrict__ batch_count, int64_t num_nodes,
                                                                 int64_t k, int replace, int64_t *__restrict__ out_indptr,
                                                                 int *__restrict__ flags, HashTable ht,
                                                                 int32_t *__restrict__ rowptr32, float *__restrict__ inv_deg) {
    __shared__ int64_t sh[16 * 12 + 5];
    const int64_t cap = (int64_t)ht.mask + 1;
    for (int64_t i = (int64_t)blockIdx.x * kRlThreads + threadIdx.x; i < cap; i += (int64_t)gridDim.x * kRlThreads) {
        ht.key[i] = kHtEmpty;
        ht.pos[i] = kHtNoPos;
    }
    if (blockIdx.x != 0) return;
    int bad = 0;
    const int64_t valid = valid_seeds(batch_count, batch);
    int64_t carry = 0;
    constexpr int PI = 12;  // entries per thread and trip (12 k: a 1024-seed second hop in one trip; strided: entry q * 1024 + t), their gathers in flight together
    for (int64_t base = 0; base <= batch; base += (int64_t)kRlThreads * PI) {  // entry `batch` (count 0) receives the total
        int64_t sd[PI], lo[PI], hi[PI], c[PI];
        // (the seed ids are requested before `valid` -- another load -- is known: one dependent round trip less)
#pragma unroll
        for (int q = 0; q < PI; ++q) {
            const int64_t i = base + (int64_t)q * kRlThreads + threadIdx.x;
            const int64_t v = batch > 0 ? node_idx[min(i, batch - 1)] : -1;
            sd[q] = i < valid ? v : -1;
        }
#pragma unroll
        for (int q = 0; q < PI; ++q) {
            const int64_t i = base + (int64_t)q * kRlThreads + threadIdx.x;
            const bool ok = sd[q] >= 0 && sd[q] < num_nodes;
            if (i < valid && !ok) bad = 1;
            // indptr[s] and indptr[s + 1] in ONE 16-byte request (8-byte aligned): this workgroup issues a random gather per
            // entry from a single CU, and that CU's request rate is what the kernel's time is made of
            // (a graph without nodes has a one-entry indptr: the unconditional 16-byte load of indptr[0..1] would read 8 bytes
            //  past it -- every seed is invalid there, nothing is loaded; the test is uniform)
            I64x2 pr{0, 0};
            if (num_nodes > 0) pr = *reinterpret_cast<const I64x2 *>(indptr + (ok ? sd[q] : 0));
            lo[q] = ok ? pr.x : 0;
            hi[q] = ok ? pr.y : 0;
        }
#pragma unroll
        for (int q = 0; q < PI; ++q) c[q] = (sd[q] >= 0 && sd[q] < num_nodes) ? count_for(hi[q] - lo[q], k, replace) : 0;
        int32_t own[PI];
#pragma unroll
        for (int q = 0; q < PI; ++q) own[q] = (int32_t)c[q];
        int64_t total;
        wg1024_strided_exclusive_scan<PI, int64_t>(c, sh, &total);
#pragma unroll
        for (int q = 0; q < PI; ++q) {
            const int64_t i = base + (int64_t)q * kRlThreads + threadIdx.x;
            if (i <= batch) out_indptr[i] = carry + c[q];
            if (rowptr32) {  // the block as the SpMM takes it (cogdl_hip_sample_adj_block): no block_prepare launch later
                if (i <= batch) rowptr32[i] = (int32_t)(carry + c[q]);
                if (i < batch && inv_deg) inv_deg[i] = own[q] > 0 ? 1.f / (float)own[q] : 0.f;
            }
        }
        carry += total;
    }
    const int any_bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) *flags = any_bad ? 1 : 0;  // (the kernels that OR further bits in run after this one)
}

// first: for every position p of Q the first position of its node (head[p]; -1 for padding), the first-occurrence flag
// and its exclusive rank inside this block of kRlTile positions (lrank[p]); blocksum[b] = first occurrences in block b.
// Positions behind the picks actually made (p >= batch + total) are padding whatever the key array holds.
struct RelabelArgs {
    const uint32_t *keys;
    const uint32_t *slots;        // [len] hash-table slot of every position's node (written by the pick kernel)
    int32_t *head, *lrank, *blocksum;
    int64_t len, batch, cap_edges;
    uint32_t pad_key;
    int padded;
    const int64_t *out_indptr_c;  // (read: total = out_indptr[batch])
    int64_t *out_indptr, *out_indices, *out_nodes, *out_edges, *out_counts;
    const int *flags;
    HashTable ht;
    int32_t *col32;               // (cogdl_hip_sample_adj_block: out_indices once more as int32; else NULL)
};

// Phases A-C of a block of ITEMS * 1024 positions starting at b0, STRIDED: item i of thread t is position
// b0 + i * 1024 + t.  For each of them the first position of its node (hp[i]; -1 for padding) and the key; hp/ky out,
// rk[i] = the exclusive rank of the position's own first-occurrence flag inside the block, *tot = first occurrences of
// the block.  Loads come in unconditional phases -- a load behind a per-lane branch is waited for at the branch's join,
// i.e. ITEMS serial round trips instead of one; indices are clamped / masked into range and what an unused position
// reads is ignored.  Two round trips to memory: {keys, slots, total}, then the gathers pos[slot].
template <int ITEMS>
__device__ __forceinline__ void relabel_block_ranks(const RelabelArgs &a, int64_t b0, int32_t (&hp)[ITEMS],
                                                    uint32_t (&ky)[ITEMS], int32_t (&rk)[ITEMS], int64_t &total,
                                                    int32_t *sh, int32_t *tot) {
    uint32_t sl[ITEMS];
    const int64_t last = a.len - 1;
    total = a.out_indptr_c[a.batch];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t pc = min(b0 + (int64_t)i * kRlThreads + threadIdx.x, last);
        ky[i] = a.keys[pc];
        sl[i] = a.slots[pc] & a.ht.mask;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) hp[i] = (int32_t)a.ht.pos[sl[i]];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t p = b0 + (int64_t)i * kRlThreads + threadIdx.x;
        const bool used = p < a.len && (p < a.batch || p - a.batch < total) && ky[i] != a.pad_key;
        if (!used) hp[i] = -1;
        rk[i] = (used && (int64_t)hp[i] == p) ? 1 : 0;
    }
    wg1024_strided_exclusive_scan<ITEMS, int32_t>(rk, sh, tot);
}

// The outputs of position p whose node has the local id `id` (h = its first position, -1: padding), and -- fixed-capacity
// form -- benign values in the unused tails: a well-formed CSR of the full capacity (see sample_pad_kernel).
__device__ __forceinline__ void relabel_emit(const RelabelArgs &a, int64_t p, int32_t h, int64_t id, uint32_t key,
                                             int64_t n_nodes, int64_t total) {
    if (h >= 0) {
        if (p >= a.batch) {
            a.out_indices[p - a.batch] = id;
            if (a.col32) a.col32[p - a.batch] = (int32_t)id;
        }
        if ((int64_t)h == p) a.out_nodes[id] = (int64_t)key;
    }
    if (a.padded) {
        if (p >= n_nodes) a.out_nodes[p] = 0;
        if (p >= a.batch) {
            const int64_t j = p - a.batch;
            if (j >= total) {
                a.out_indices[j] = 0;
                a.out_edges[j] = 0;
                if (a.col32) a.col32[j] = 0;
            }
            a.out_indptr[a.batch + 1 + j] = total;
        }
    }
}

// first (several blocks): head[p], lrank[p] (the rank of p's OWN first-occurrence flag inside its block), blocksum[b].
__global__ __launch_bounds__(kRlThreads) void sample_first_kernel(const RelabelArgs a) {
    __shared__ int32_t sh[16 * kRlItems + 5];
    const int64_t b0 = (int64_t)blockIdx.x * kRlTile;
    int32_t hp[kRlItems], rk[kRlItems], tot;
    uint32_t ky[kRlItems];
    int64_t total;
    relabel_block_ranks<kRlItems>(a, b0, hp, ky, rk, total, sh, &tot);
#pragma unroll
    for (int i = 0; i < kRlItems; ++i) {
        const int64_t p = b0 + (int64_t)i * kRlThreads + threadIdx.x;
        if (p < a.len) {
            a.head[p] = hp[i];
            a.lrank[p] = rk[i];
        }
    }
    if (threadIdx.x == 0) a.blocksum[blockIdx.x] = tot;
}

// relabel (several blocks): local id of position p = (first occurrences in the blocks before head[p]'s) + lrank[head[p]].
__global__ __launch_bounds__(kRlThreads) void sample_relabel_hash_kernel(const RelabelArgs a, int64_t n_blocks) {
    __shared__ int32_t sh[kRlThreads / kWave + 1];
    __shared__ int32_t bprefix[kRlMaxBlocks];
    const int64_t b0 = (int64_t)blockIdx.x * kRlTile;
    const int64_t last = a.len - 1;
    // everything this workgroup reads that does not depend on another read is requested first
    int32_t hd[kRlItems], lr[kRlItems];
    uint32_t ky[kRlItems];
    const int64_t total = a.out_indptr_c[a.batch];
    const int fl = *a.flags;
#pragma unroll
    for (int i = 0; i < kRlItems; ++i) {
        const int64_t pc = min(b0 + (int64_t)i * kRlThreads + threadIdx.x, last);
        hd[i] = a.head[pc];
        ky[i] = a.keys[pc];
    }
    // exclusive prefix of the block sums, by every workgroup for itself (n_blocks <= kRlMaxBlocks)
    int32_t carry = 0;
    for (int64_t base = 0; base < n_blocks; base += kRlThreads) {
        const int64_t i = base + threadIdx.x;
        const int32_t v = i < n_blocks ? a.blocksum[i] : 0;
        int32_t tot;
        const int32_t ex = wg1024_exclusive_scan<int32_t>(v, sh, &tot);
        if (i < n_blocks) bprefix[i] = carry + ex;
        carry += tot;
    }
#pragma unroll
    for (int i = 0; i < kRlItems; ++i) {
        if (b0 + (int64_t)i * kRlThreads + threadIdx.x > last) hd[i] = -1;
        lr[i] = a.lrank[max(hd[i], 0)];  // (independent gathers, all in flight)
    }
    __syncthreads();
    const int64_t n_nodes = carry;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.out_counts[0] = n_nodes;
        a.out_counts[1] = total;
        a.out_counts[2] = fl;  // 0, or why the result is invalid
    }
#pragma unroll
    for (int i = 0; i < kRlItems; ++i) {
        const int64_t p = b0 + (int64_t)i * kRlThreads + threadIdx.x;
        if (p <= last) {
            const int64_t id = hd[i] >= 0 ? (int64_t)bprefix[hd[i] / kRlTile] + lr[i] : 0;
            relabel_emit(a, p, hd[i], id, ky[i], n_nodes, total);
        }
    }
}

// |Q| <= kRlTile1: ONE workgroup, one launch, two round trips to memory: the ranks of the block travel through LDS
// (16-bit: ranks < kRlTile1), nothing is re-read from global memory.
// ITEMS positions per thread (|Q| <= 1024 ITEMS): the first hop of a 128-seed step has 1408 positions, not 16384 -- one
// workgroup's gathers leave from one CU, so the kernel's time is its number of requests.
template <int ITEMS>
__global__ __launch_bounds__(kRlThreads) void sample_relabel_single_kernel(const RelabelArgs a) {
    static_assert(ITEMS <= kRlItems1, "at most kRlTile1 positions");
    __shared__ int32_t sh[16 * ITEMS + 5];
    __shared__ uint16_t lrank16[kRlThreads * ITEMS];
    const int fl = *a.flags;
    int32_t hp[ITEMS], rk[ITEMS], tot;
    uint32_t ky[ITEMS];
    int64_t total;
    relabel_block_ranks<ITEMS>(a, 0, hp, ky, rk, total, sh, &tot);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) lrank16[i * kRlThreads + threadIdx.x] = (uint16_t)rk[i];
    __syncthreads();
    const int64_t n_nodes = tot;
    if (threadIdx.x == 0) {
        a.out_counts[0] = n_nodes;
        a.out_counts[1] = total;
        a.out_counts[2] = fl;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t p = (int64_t)i * kRlThreads + threadIdx.x;
        if (p < a.len) {
            const int64_t id = hp[i] >= 0 ? (int64_t)lrank16[hp[i]] : 0;
            relabel_emit(a, p, hp[i], id, ky[i], n_nodes, total);
        }
    }
}

__global__ void sample_counts_kernel(const int64_t *__restrict__ indptr, const int64_t *__restrict__ node_idx,
                                     int64_t batch, const int64_t *__restrict__ batch_count, int64_t num_nodes, int64_t k,
                                     int replace, int32_t *__restrict__ cnt, int *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > batch) return;
    int32_t c = 0;
    if (i < valid_seeds(batch_count, batch)) {
        const int64_t s = node_idx[i];
        if (s < 0 || s >= num_nodes) atomicOr(flags, 1);
        else c = (int32_t)count_for(indptr[s + 1] - indptr[s], k, replace);
    }
    cnt[i] = c;  // cnt[batch] = 0: the scan then also yields the total
}

__global__ __launch_bounds__(256) void sample_pick_kernel(const int64_t *__restrict__ indptr,
                                                          const int64_t *__restrict__ indices,
                                                          const int64_t *__restrict__ node_idx, int64_t batch,
                                                          const int64_t *__restrict__ batch_count,
                                                          int64_t num_nodes, int64_t k, int replace, uint64_t seed,
                                                          const uint64_t *__restrict__ seed_dev,
                                                          const int64_t *__restrict__ out_indptr,
                                                          int64_t *__restrict__ out_edges, uint32_t *__restrict__ keys,
                                                          uint32_t pad_key, int64_t cap_edges, int *__restrict__ flags,
                                                          HashTable ht, uint32_t *__restrict__ slots) {
    __shared__ int32_t chosen_all[4][kSampleMaxK];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= batch) return;
    // everything that does not depend on another load is requested first (each dependent round trip is ~2 us of a chain
    // of five): the seed id and its output offsets next to the count of seeds in use and the RNG seed word
    const int64_t s = node_idx[i];
    const int64_t off = out_indptr[i], cnt = out_indptr[i + 1] - off;
    const uint64_t seed_add = seed_dev ? *seed_dev : 0;  // (a captured graph re-reads it on every replay; `seed` then is a per-call offset)
    if (i >= valid_seeds(batch_count, batch)) {  // an unused seed slot: sorts last, gets no local id
        if (lane == 0) keys[i] = pad_key;
        return;
    }
    seed += seed_add;
    if (s < 0 || s >= num_nodes) {  // flagged by the count kernel; its count is 0
        if (lane == 0) keys[i] = pad_key;
        return;
    }
    if (lane == 0) {
        keys[i] = (uint32_t)s;
        if (ht.key) slots[i] = ht_insert(ht, (uint32_t)s, (uint32_t)i);
    }
    const I64x2 pr = *reinterpret_cast<const I64x2 *>(indptr + s);
    const int64_t start = pr.x, deg = pr.y - start;
    auto emit = [&](int64_t j, int64_t pos) {
        if (off + j >= cap_edges) {
            atomicOr(flags, 4);
            return;
        }
        const int64_t nb = indices[pos];
        const bool bad = nb < 0 || nb >= num_nodes;
        if (bad) atomicOr(flags, 2);
        out_edges[off + j] = pos;
        keys[batch + off + j] = bad ? pad_key : (uint32_t)nb;  // (an id outside the graph gets no local id; flags say so)
        if (ht.key && !bad) slots[batch + off + j] = ht_insert(ht, (uint32_t)nb, (uint32_t)(batch + off + j));
    };
    if (k < 0 || (!replace && deg <= k)) {  // the whole row, in CSR order
        for (int64_t j = lane; j < cnt; j += kWave) emit(j, start + j);
    } else if (replace) {
        for (int64_t j = lane; j < cnt; j += kWave) emit(j, start + (int64_t)(draw(seed, (uint64_t)i, (uint64_t)j) % (uint64_t)deg));
    } else {
        // Floyd: for jj = deg-k .. deg-1 draw t in [0, jj]; take t unless already chosen, then take jj.  Uniform over
        // the k-subsets.  The loop is wave-uniform; the set lives in LDS, one ballot per draw tests membership.
        volatile int32_t *chosen = chosen_all[wave];
        const int kk = (int)k;
        int n = 0;
        for (int64_t jj = deg - kk; jj < deg; ++jj) {
            const int32_t t = (int32_t)(draw(seed, (uint64_t)i, (uint64_t)jj) % (uint64_t)(jj + 1));
            bool hit = false;
            for (int r = lane; r < n; r += kWave) hit |= (chosen[r] == t);
            const int32_t pick = __any(hit) ? (int32_t)jj : t;
            if (lane == 0) chosen[n] = pick;
            ++n;
            __builtin_amdgcn_wave_barrier();  // LDS operations of one wave complete in order: the write is visible
        }
        for (int r = lane; r < kk; r += kWave) {  // ascending CSR position: rank by counting (the values are distinct)
            const int32_t v = chosen[r];
            int rank = 0;
            for (int q = 0; q < kk; ++q) rank += (chosen[q] < v) ? 1 : 0;
            emit(rank, start + v);
        }
    }
}

// Unused tail of the sequence (capacity > actual number of picks): a key above every node id, so it sorts last.
// `padded`: the unused tails of the outputs get benign values too -- out_indices / out_edges 0 (a valid id / position),
// out_indptr[batch+1 .. batch+cap_edges] = E' (rows without edges) -- so that the block is a well-formed CSR of the
// full capacity whatever was sampled.
__global__ void sample_pad_kernel(int64_t *__restrict__ out_indptr, int64_t batch, int64_t cap_edges,
                                  uint32_t *__restrict__ keys, uint32_t pad_key, int padded,
                                  int64_t *__restrict__ out_indices, int64_t *__restrict__ out_edges) {
    const int64_t total = out_indptr[batch];
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = total + t0; j < cap_edges; j += stride) {
        keys[batch + j] = pad_key;
        if (padded) {
            out_indices[j] = 0;
            out_edges[j] = 0;
        }
    }
    if (padded)
        for (int64_t j = t0; j < cap_edges; j += stride) out_indptr[batch + 1 + j] = total;
}

// sorted order j: first[pos[j]] = is this the first occurrence of its node in Q; head[j] = j for run heads, else 0
__global__ void sample_mark_kernel(const uint32_t *__restrict__ skeys, const int32_t *__restrict__ spos, int64_t len,
                                   uint32_t pad_key, int32_t *__restrict__ first, int32_t *__restrict__ head) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= len) return;
    const uint32_t key = skeys[j];
    const bool is_head = key != pad_key && (j == 0 || skeys[j - 1] != key);
    first[spos[j]] = is_head ? 1 : 0;  // the sort is stable: a run's head is its smallest position in Q
    head[j] = is_head ? (int32_t)j : 0;
}

__global__ void sample_relabel_kernel(const uint32_t *__restrict__ skeys, const int32_t *__restrict__ spos,
                                      const int32_t *__restrict__ first, const int32_t *__restrict__ rank,
                                      const int32_t *__restrict__ head_of, int64_t len, int64_t batch,
                                      uint32_t pad_key, const int64_t *__restrict__ out_indptr,
                                      int64_t *__restrict__ out_indices, int64_t *__restrict__ out_nodes,
                                      int64_t *__restrict__ out_counts, const int *__restrict__ flags, int padded) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (padded && j < len && j >= (int64_t)rank[len - 1] + first[len - 1]) out_nodes[j] = 0;  // unused slots: a valid id
    if (j == 0) {
        out_counts[0] = (int64_t)rank[len - 1] + first[len - 1];  // nodes
        out_counts[1] = out_indptr[batch];                        // edges
        out_counts[2] = *flags;                                   // 0, or why the result is invalid
    }
    if (j >= len) return;
    const uint32_t key = skeys[j];
    if (key == pad_key) return;
    const int32_t hj = head_of[j];
    const int64_t id = rank[spos[hj]];
    const int32_t p = spos[j];
    if (p >= batch) out_indices[p - batch] = id;
    if (hj == (int32_t)j) out_nodes[id] = (int64_t)key;
}


struct SampleWs {
    int32_t *cnt;                  // [batch + 1]
    uint32_t *keys, *skeys;        // [len]
    int32_t *spos, *first, *rank, *head, *head_of;  // [len]
    int *flags;
    char *temp;
    size_t temp_bytes, total;
    uint32_t *tkey, *tpos;         // hash table of the relabelling, `tcap` slots each
    int64_t tcap, n_blocks;
    int32_t *blocksum;             // [n_blocks]
};

static SampleWs carve(void *base, int64_t batch, int64_t cap_edges, int64_t num_nodes) {
    const int64_t len = batch + cap_edges;
    SampleWs w{};
    // (sort-based relabelling: keys are the node ids plus the padding key num_nodes: num_nodes + 1 key values)
    const size_t sort_t = sort_positions_workspace_bytes(num_nodes + 1, std::max<int64_t>(len, 1));
    const size_t scan_t = device_scan_temp_bytes(std::max<int64_t>(len, 1), sizeof(int32_t));
    const size_t scan64_t = device_scan_temp_bytes(batch + 1, sizeof(int64_t));
    w.temp_bytes = std::max(sort_t, std::max(scan_t, scan64_t));
    char *p = (char *)base;
    auto take = [&](size_t bytes) {
        char *q = p;
        p += align256(bytes);
        return q;
    };
    w.flags = (int *)take(256);
    w.cnt = (int32_t *)take((size_t)(batch + 1) * 4);
    w.keys = (uint32_t *)take((size_t)len * 4);
    w.skeys = (uint32_t *)take((size_t)len * 4);
    w.spos = (int32_t *)take((size_t)len * 4);
    w.first = (int32_t *)take((size_t)len * 4);
    w.rank = (int32_t *)take((size_t)len * 4);
    w.head = (int32_t *)take((size_t)len * 4);
    w.head_of = (int32_t *)take((size_t)len * 4);
    w.temp = take(w.temp_bytes);
    w.n_blocks = len <= kRlTile1 ? 1 : (len + kRlTile - 1) / kRlTile;
    w.tcap = 2048;
    while (w.tcap < 2 * len) w.tcap <<= 1;
    if (w.n_blocks > kRlMaxBlocks) w.tcap = 0;  // the sort-based relabelling takes these sizes: no table (it would be GBs)
    w.tkey = (uint32_t *)take((size_t)w.tcap * 4);
    w.tpos = (uint32_t *)take((size_t)w.tcap * 4);
    w.blocksum = (int32_t *)take((size_t)std::min<int64_t>(w.n_blocks, kRlMaxBlocks) * 4);
    w.total = (size_t)(p - (char *)base);
    return w;
}

}  // namespace cogdl

using namespace cogdl;

extern "C" int cogdl_hip_block_prepare(const int64_t *row_ptr, const int64_t *col, int64_t n_rows, int64_t n_slots,
                                       int32_t *rowptr32, int32_t *col32, float *inv_deg, void *stream);

extern "C" size_t cogdl_hip_sample_adj_workspace_bytes(int64_t batch, int64_t cap_edges, int64_t num_nodes) {
    if (batch < 0 || cap_edges < 0 || num_nodes < 0) return 0;
    return carve(nullptr, batch, cap_edges, num_nodes).total;
}

static int sample_adj_impl(const int64_t *indptr, const int64_t *indices, int64_t num_nodes, const int64_t *node_idx,
                           int64_t batch, const int64_t *batch_count, int64_t num_neighbors, int replace, uint64_t seed,
                           const uint64_t *seed_dev, int padded, int64_t *out_indptr, int64_t *out_indices,
                           int64_t *out_nodes, int64_t *out_edges, int64_t cap_edges, int64_t *out_counts,
                           void *workspace, size_t workspace_bytes, void *stream, int32_t *rowptr32 = nullptr,
                           int32_t *col32 = nullptr, float *inv_deg = nullptr) {
    if (batch < 0 || cap_edges < 0 || num_nodes < 0 || !out_indptr || !out_counts || !workspace) return COGDL_HIP_EINVAL;
    if (batch > 0 && (!indptr || !node_idx || !out_nodes)) return COGDL_HIP_EINVAL;
    if (cap_edges > 0 && (!indices || !out_indices || !out_edges)) return COGDL_HIP_EINVAL;
    if (num_nodes > (int64_t(1) << 31) || batch + cap_edges > 0x7fffffff) return COGDL_HIP_ERANGE;
    if (!replace && num_neighbors > kSampleMaxK) return COGDL_HIP_ERANGE;
    if (!aligned_to(workspace, 256)) return COGDL_HIP_EALIGN;
    const SampleWs w = carve(workspace, batch, cap_edges, num_nodes);
    if (workspace_bytes < w.total) return COGDL_HIP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t len = batch + cap_edges;
    const uint32_t pad_key = 1u << sample_key_bits(num_nodes);
    auto fail = [](hipError_t e) {
        g_last_hip_error = (int)e;
        return COGDL_HIP_ELAUNCH;
    };
    // (beyond 6.7e7 positions -- the block prefixes no longer fit one workgroup's LDS scan -- the sort-based form below
    //  takes over: it has no such limit)
    const bool hash_relabel = g_tuning[kTuneSampleRelabel] == 0 && w.n_blocks <= kRlMaxBlocks && w.tcap > 0;
    if (hash_relabel) {
        HashTable ht{w.tkey, w.tpos, (uint32_t)(w.tcap - 1)};
        const unsigned prep_blocks = (unsigned)std::min<int64_t>(256, std::max<int64_t>(1, w.tcap / (kRlThreads * 8)));
        hipLaunchKernelGGL(sample_prep_kernel, dim3(prep_blocks), dim3(kRlThreads), 0, s, indptr, node_idx, batch, batch_count,
                           num_nodes, num_neighbors, replace, out_indptr, w.flags, ht, rowptr32, inv_deg);
        if (len == 0) {
            hipError_t e0 = fill_u32_async(out_counts, 0u, 6, s);
            return e0 == hipSuccess ? launch_status() : fail(e0);
        }
        if (batch > 0)
            hipLaunchKernelGGL(sample_pick_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, s, indptr, indices,
                               node_idx, batch, batch_count, num_nodes, num_neighbors, replace, seed, seed_dev, out_indptr,
                               out_edges, w.keys, pad_key, cap_edges, w.flags, ht, (uint32_t *)w.first);
        RelabelArgs a{};
        a.keys = w.keys;
        a.slots = (const uint32_t *)w.first;
        a.head = w.head_of;
        a.lrank = w.rank;
        a.blocksum = w.blocksum;
        a.len = len;
        a.batch = batch;
        a.cap_edges = cap_edges;
        a.pad_key = pad_key;
        a.padded = padded;
        a.out_indptr_c = out_indptr;
        a.out_indptr = out_indptr;
        a.out_indices = out_indices;
        a.out_nodes = out_nodes;
        a.out_edges = out_edges;
        a.out_counts = out_counts;
        a.flags = w.flags;
        a.ht = ht;
        a.col32 = col32;
        if (w.n_blocks == 1) {
            if (len <= 2 * kRlThreads) hipLaunchKernelGGL(sample_relabel_single_kernel<2>, dim3(1), dim3(kRlThreads), 0, s, a);
            else if (len <= 4 * kRlThreads) hipLaunchKernelGGL(sample_relabel_single_kernel<4>, dim3(1), dim3(kRlThreads), 0, s, a);
            else if (len <= 8 * kRlThreads) hipLaunchKernelGGL(sample_relabel_single_kernel<8>, dim3(1), dim3(kRlThreads), 0, s, a);
            else if (len <= 12 * kRlThreads) hipLaunchKernelGGL(sample_relabel_single_kernel<12>, dim3(1), dim3(kRlThreads), 0, s, a);
            else hipLaunchKernelGGL(sample_relabel_single_kernel<kRlItems1>, dim3(1), dim3(kRlThreads), 0, s, a);
        } else {
            hipLaunchKernelGGL(sample_first_kernel, dim3((unsigned)w.n_blocks), dim3(kRlThreads), 0, s, a);
            hipLaunchKernelGGL(sample_relabel_hash_kernel, dim3((unsigned)w.n_blocks), dim3(kRlThreads), 0, s, a, w.n_blocks);
        }
        return launch_status();
    }
    hipError_t e = fill_u32_async(w.flags, 0u, 1, s);
    if (e != hipSuccess) return fail(e);
    hipLaunchKernelGGL(sample_counts_kernel, dim3((unsigned)((batch + 256) / 256)), dim3(256), 0, s, indptr, node_idx, batch,
                       batch_count, num_nodes, num_neighbors, replace, w.cnt, w.flags);
    // The sort-based form pads with the key num_nodes (the hash form's power of two would make 2^31 + 1 key values of a
    // graph of more than 2^30 nodes -- beyond what the position sort takes): ids 0 .. num_nodes - 1 sort
    // in front of it, the sort sees num_nodes + 1 key values.
    if (num_nodes + 1 >= 0x7fffffff) return COGDL_HIP_ERANGE;
    const uint32_t pad_sort = (uint32_t)num_nodes;
    int rc_scan = device_exclusive_sum((const int32_t *)w.cnt, out_indptr, batch + 1, w.temp, s);
    if (rc_scan != COGDL_HIP_OK) return rc_scan;
    if (len == 0) {
        e = fill_u32_async(out_counts, 0u, 6, s);
        if (e != hipSuccess) return fail(e);
        return rowptr32 ? cogdl_hip_block_prepare(out_indptr, out_indices, 0, 0, rowptr32, col32, inv_deg, stream) : launch_status();
    }
    if (batch > 0)
        hipLaunchKernelGGL(sample_pick_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, s, indptr, indices, node_idx,
                           batch, batch_count, num_nodes, num_neighbors, replace, seed, seed_dev, out_indptr, out_edges, w.keys,
                           pad_sort, cap_edges, w.flags, HashTable{nullptr, nullptr, 0u}, nullptr);
    if (cap_edges > 0)
        hipLaunchKernelGGL(sample_pad_kernel, dim3((unsigned)std::min<int64_t>((cap_edges + 255) / 256, 1024)), dim3(256), 0, s,
                           out_indptr, batch, cap_edges, w.keys, pad_sort, padded, out_indices, out_edges);
    {
        const int rc = sort_positions_by_key(w.keys, len, num_nodes + 1, w.spos, w.temp, w.temp_bytes, s);
        if (rc != COGDL_HIP_OK) return rc;
        hipLaunchKernelGGL(sample_gather_keys_kernel, dim3((unsigned)std::min<int64_t>((len + 255) / 256, 1 << 16)), dim3(256), 0, s,
                           w.keys, w.spos, w.skeys, len);
    }
    const unsigned blocks = (unsigned)((len + 255) / 256);
    hipLaunchKernelGGL(sample_mark_kernel, dim3(blocks), dim3(256), 0, s, w.skeys, w.spos, len, pad_sort, w.first, w.head);
    rc_scan = device_exclusive_sum((const int32_t *)w.first, w.rank, len, w.temp, s);
    if (rc_scan != COGDL_HIP_OK) return rc_scan;
    // (head[j] = j at the head of a run of equal keys, else 0: the running maximum is the head of the run j sits in)
    rc_scan = device_scan<true>((const int32_t *)w.head, w.head_of, len, int32_t(0), int32_t(0), ScanMax(), w.temp, s);
    if (rc_scan != COGDL_HIP_OK) return rc_scan;
    hipLaunchKernelGGL(sample_relabel_kernel, dim3(blocks), dim3(256), 0, s, w.skeys, w.spos, w.first, w.rank, w.head_of, len,
                       batch, pad_sort, out_indptr, out_indices, out_nodes, out_counts, w.flags, padded);
    if (rowptr32)  // (the sort-based form does not write the int32 block itself)
        return cogdl_hip_block_prepare(out_indptr, out_indices, batch, cap_edges, rowptr32, col32, inv_deg, stream);
    return launch_status();
}

extern "C" int cogdl_hip_sample_adj(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                    const int64_t *node_idx, int64_t batch, int64_t num_neighbors, int replace,
                                    uint64_t seed, int64_t *out_indptr, int64_t *out_indices, int64_t *out_nodes,
                                    int64_t *out_edges, int64_t cap_edges, int64_t *out_counts, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return sample_adj_impl(indptr, indices, num_nodes, node_idx, batch, nullptr, num_neighbors, replace, seed, nullptr, 0,
                           out_indptr, out_indices, out_nodes, out_edges, cap_edges, out_counts, workspace,
                           workspace_bytes, stream);
}

extern "C" int cogdl_hip_sample_adj_padded(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                           const int64_t *node_idx, int64_t batch, const int64_t *batch_count,
                                           int64_t num_neighbors, int replace, uint64_t seed, const uint64_t *seed_dev,
                                           int64_t *out_indptr, int64_t *out_indices, int64_t *out_nodes,
                                           int64_t *out_edges, int64_t cap_edges, int64_t *out_counts, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    return sample_adj_impl(indptr, indices, num_nodes, node_idx, batch, batch_count, num_neighbors, replace, seed, seed_dev,
                           1, out_indptr, out_indices, out_nodes, out_edges, cap_edges, out_counts, workspace,
                           workspace_bytes, stream);
}

extern "C" int cogdl_hip_sample_adj_block(const int64_t *indptr, const int64_t *indices, int64_t num_nodes,
                                          const int64_t *node_idx, int64_t batch, const int64_t *batch_count,
                                          int64_t num_neighbors, int replace, uint64_t seed, const uint64_t *seed_dev,
                                          int64_t *out_indptr, int64_t *out_indices, int64_t *out_nodes,
                                          int64_t *out_edges, int64_t cap_edges, int64_t *out_counts, int32_t *rowptr32,
                                          int32_t *col32, float *inv_deg, void *workspace, size_t workspace_bytes,
                                          void *stream) {
    if (!rowptr32 || (cap_edges > 0 && !col32)) return COGDL_HIP_EINVAL;
    return sample_adj_impl(indptr, indices, num_nodes, node_idx, batch, batch_count, num_neighbors, replace, seed, seed_dev,
                           1, out_indptr, out_indices, out_nodes, out_edges, cap_edges, out_counts, workspace,
                           workspace_bytes, stream, rowptr32, col32, inv_deg);
}
