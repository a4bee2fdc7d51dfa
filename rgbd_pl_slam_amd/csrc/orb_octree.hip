// orb_octree.hip -- ORBextractor::DistributeOctTree + ExtractorNode::DivideNode on the GPU
// (include/ORBextractor.h:91,37; so@0x73c60, so@0x70c60).
//
// The reference algorithm is a sequential std::list quad-tree.  It is restated here as a
// level-synchronous, data-parallel program that produces the SAME node list in the SAME order
// (model: tests/models/octree_parallel.py, verified order-exact against the reference binary):
//   * a key never moves in memory; it only carries the list position of its node (node_of);
//   * one "sweep" divides a set of nodes: per-key quadrant + LDS atomic histogram, per-node child
//     rectangles, block scans that give every child / surviving node its position in the new list
//     (children of the LAST divided node first, inside a node n4..n1, then the untouched nodes);
//   * phase 2 ("largest first until N nodes") sorts the expandable nodes by (key count, creation
//     rank) with a bitonic sort, divides ALL of them speculatively, and keeps the prefix the
//     sequential loop would have processed before its `break`;
//   * the kept keypoint of a node is a 64-bit atomicMax over (response, ~key index).
// One 256-thread workgroup per (level, frame); node state lives in LDS, keys in global scratch.
#include "plf_common.h"
#include "orb_geom.h"

struct OctArrays {
    int2 *rectA, *rectB;   // (x0 | y0 << 16, x1 | y1 << 16)
    int *nA, *nB;          // keys per node
    int *slot_of;          // per old node: slot in the work list or -1
    int *newpos;           // per old node: position in the new list when it survives
    int *work;             // node positions in processing order
    int *mid;              // per slot: mx | my << 16
    int *ccnt;             // per slot: 4 child key counts
    int *s1, *s2;          // per slot scratch (scans)
    int *cand;             // expandable nodes (positions), creation order
    unsigned long long *skey;  // sort keys / best-key words
    int *scan_tmp;         // 257
};

__device__ __forceinline__ int quadrant(float kx, float ky, int mx, int my)
{
    if (kx < (float)mx) return (ky < (float)my) ? 0 : 2;
    return (ky < (float)my) ? 1 : 3;
}

__device__ __forceinline__ int2 child_rect(int2 p, int mid, int q)
{
    const int x0 = p.x & 0xFFFF, y0 = (unsigned)p.x >> 16, x1 = p.y & 0xFFFF, y1 = (unsigned)p.y >> 16;
    const int mx = mid & 0xFFFF, my = (unsigned)mid >> 16;
    int ax0, ay0, ax1, ay1;
    if (q == 0) { ax0 = x0; ay0 = y0; ax1 = mx; ay1 = my; }
    else if (q == 1) { ax0 = mx; ay0 = y0; ax1 = x1; ay1 = my; }
    else if (q == 2) { ax0 = x0; ay0 = my; ax1 = mx; ay1 = y1; }
    else { ax0 = mx; ay0 = my; ax1 = x1; ay1 = y1; }
    return make_int2(ax0 | (ay0 << 16), ax1 | (ay1 << 16));
}

// Divide the nodes work[0..S).  If limitN >= 0 (phase 2) only the prefix the sequential loop would
// reach before `if (size >= N) break` is really divided.  Returns the new list length; *ncand_out =
// number of new expandable children (written to A.cand in creation order).
__device__ int octree_divide(OctArrays &A, int Lsz, int S, int limitN, const uint2 *keys, int *node_of, uint8_t *quad, int nk,
                             int *ncand_out)
{
    const int T = blockDim.x, t = threadIdx.x;
    for (int i = t; i < Lsz; i += T) A.slot_of[i] = -1;
    __syncthreads();
    for (int s = t; s < S; s += T) {
        const int pos = A.work[s];
        A.slot_of[pos] = s;
        const int2 r = A.rectA[pos];
        const int x0 = r.x & 0xFFFF, y0 = (unsigned)r.x >> 16, x1 = r.y & 0xFFFF, y1 = (unsigned)r.y >> 16;
        const int halfX = (int)ceilf((float)(x1 - x0) / 2.0f), halfY = (int)ceilf((float)(y1 - y0) / 2.0f);
        A.mid[s] = (x0 + halfX) | ((y0 + halfY) << 16);
        A.ccnt[4 * s] = A.ccnt[4 * s + 1] = A.ccnt[4 * s + 2] = A.ccnt[4 * s + 3] = 0;
    }
    __syncthreads();
    for (int k = t; k < nk; k += T) {
        const int s = A.slot_of[node_of[k]];
        if (s >= 0) {
            const uint2 kv = keys[k];
            const int m = A.mid[s];
            const int q = quadrant((float)(kv.x & 0xFFFF), (float)(kv.x >> 16), m & 0xFFFF, (unsigned)m >> 16);
            atomicAdd(&A.ccnt[4 * s + q], 1);
            quad[k] = (uint8_t)q;
        }
    }
    __syncthreads();
    // children per slot; phase 2: find how many slots are really processed
    for (int s = t; s < S; s += T) {
        const int c = (A.ccnt[4 * s] > 0) + (A.ccnt[4 * s + 1] > 0) + (A.ccnt[4 * s + 2] > 0) + (A.ccnt[4 * s + 3] > 0);
        A.s1[s] = c;
    }
    __syncthreads();
    int P = S;
    if (limitN >= 0) {
        // size after processing slots 0..s = Lsz + sum_{s'<=s} (c - 1); first s with size >= N ends the loop
        for (int s = t; s < S; s += T) A.s2[s] = A.s1[s] - 1;
        __syncthreads();
        plf_block_excl_scan(A.s2, S, A.scan_tmp);
        if (t == 0) A.scan_tmp[0] = S;
        __syncthreads();
        for (int s = t; s < S; s += T)
            if (Lsz + A.s2[s] + A.s1[s] - 1 >= limitN) atomicMin(&A.scan_tmp[0], s + 1);
        __syncthreads();
        P = A.scan_tmp[0];
        __syncthreads();
        for (int s = P + t; s < S; s += T) A.slot_of[A.work[s]] = -1;  // not divided: nodes stay
        __syncthreads();
    }
    // after[s] = children of slots s+1..P-1 (they end up in FRONT of slot s's children)
    for (int s = t; s < P; s += T) A.s2[s] = A.s1[s];
    __syncthreads();
    const int total_children = plf_block_excl_scan(A.s2, P, A.scan_tmp);  // s2 = exclusive prefix
    for (int s = t; s < P; s += T) A.s2[s] = total_children - A.s2[s] - A.s1[s];  // = after[s]
    // surviving nodes keep their relative order behind the children
    for (int i = t; i < Lsz; i += T) A.newpos[i] = (A.slot_of[i] < 0) ? 1 : 0;
    __syncthreads();
    const int nstay = plf_block_excl_scan(A.newpos, Lsz, A.scan_tmp);
    for (int i = t; i < Lsz; i += T) {
        if (A.slot_of[i] < 0) {
            const int np = total_children + A.newpos[i];
            A.newpos[i] = np;
            A.rectB[np] = A.rectA[i];
            A.nB[np] = A.nA[i];
        }
    }
    // children: position = after[s] + (number of non-empty children with a larger index)
    for (int s = t; s < P; s += T) {
        const int2 pr = A.rectA[A.work[s]];
        const int m = A.mid[s];
        int rank = 0, nbig = 0;
        for (int q = 3; q >= 0; q--) {
            const int c = A.ccnt[4 * s + q];
            if (c > 0) {
                const int np = A.s2[s] + rank;
                A.rectB[np] = child_rect(pr, m, q);
                A.nB[np] = c;
                A.ccnt[4 * s + q] = (int)((unsigned)c | ((unsigned)np << 20));  // remember the child's position (c < 2^20 keys, np < 2^12)
                rank++;
                if (c > 1) nbig++;
            }
        }
        A.s1[s] = nbig;
    }
    __syncthreads();
    // keys follow their node
    for (int k = t; k < nk; k += T) {
        const int nd = node_of[k];
        const int s = A.slot_of[nd];
        node_of[k] = (s >= 0) ? (int)((unsigned)A.ccnt[4 * s + quad[k]] >> 20) : A.newpos[nd];
    }
    // expandable children in creation order (slot ascending, n1..n4)
    const int ncand = plf_block_excl_scan(A.s1, P, A.scan_tmp);
    for (int s = t; s < P; s += T) {
        int o = A.s1[s];
        for (int q = 0; q < 4; q++) {
            const int v = A.ccnt[4 * s + q];
            if ((v & 0xFFFFF) > 1) A.cand[o++] = (int)((unsigned)v >> 20);
        }
    }
    __syncthreads();
    // swap buffers
    int2 *tr = A.rectA; A.rectA = A.rectB; A.rectB = tr;
    int *tn = A.nA; A.nA = A.nB; A.nB = tn;
    *ncand_out = ncand;
    return total_children + nstay;
}

__global__ void __launch_bounds__(256) k_octree(const int2 *__restrict__ cellinfo, const uint2 *__restrict__ pool,
                                                int *__restrict__ celloff, uint2 *__restrict__ keys_all,
                                                int *__restrict__ nodeof_all, uint8_t *__restrict__ quad_all,
                                                uint2 *__restrict__ sel, int *__restrict__ selcnt, int *__restrict__ ncand_dbg,
                                                int *__restrict__ status, OrbGeom g, int cap_nodes, int cap_sort)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = blockIdx.x, f = blockIdx.y, T = blockDim.x, t = threadIdx.x;
    const OrbLevel &L = g.lv[l];
    OctArrays A;
    {
        char *p = smem;
        A.skey = (unsigned long long *)p; p += sizeof(unsigned long long) * cap_sort;
        A.rectA = (int2 *)p; p += sizeof(int2) * cap_nodes;
        A.rectB = (int2 *)p; p += sizeof(int2) * cap_nodes;
        A.nA = (int *)p; p += sizeof(int) * cap_nodes;
        A.nB = (int *)p; p += sizeof(int) * cap_nodes;
        A.slot_of = (int *)p; p += sizeof(int) * cap_nodes;
        A.newpos = (int *)p; p += sizeof(int) * cap_nodes;
        A.work = (int *)p; p += sizeof(int) * cap_nodes;
        A.mid = (int *)p; p += sizeof(int) * cap_nodes;
        A.ccnt = (int *)p; p += sizeof(int) * 4 * cap_nodes;
        A.s1 = (int *)p; p += sizeof(int) * cap_nodes;
        A.s2 = (int *)p; p += sizeof(int) * cap_nodes;
        A.cand = (int *)p; p += sizeof(int) * cap_nodes;
        A.scan_tmp = (int *)p;
    }
    // ---- gather this level's candidates in reference order (cell-major, raster inside a cell)
    const int2 *ci = cellinfo + (size_t)f * g.cells_total + L.cell_base;
    int *off = celloff + (size_t)f * g.cells_total + L.cell_base;
    for (int c = t; c < L.ncells; c += T) off[c] = ci[c].y;
    __syncthreads();
    const int nk = plf_block_excl_scan(off, L.ncells, A.scan_tmp);
    uint2 *keys = keys_all + (size_t)f * g.pool_stride + L.pool_off;
    int *node_of = nodeof_all + (size_t)f * g.pool_stride + L.pool_off;
    uint8_t *quad = quad_all + (size_t)f * g.pool_stride + L.pool_off;
    const uint2 *pl = pool + (size_t)f * g.pool_stride + L.pool_off;
    if (t == 0) ncand_dbg[f * g.nlevels + l] = nk;
    if (nk == 0) {
        if (t == 0) selcnt[f * g.nlevels + l] = 0;
        return;
    }
    for (int c = t; c < L.ncells; c += T) {
        const int2 bc = ci[c];
        const int o = off[c];
        for (int i = 0; i < bc.y; i++) keys[o + i] = pl[bc.x + i];
    }
    __syncthreads();
    // ---- roots
    const int W = L.w - 2 * PLF_MINB, H = L.h - 2 * PLF_MINB, N = L.quota;
    const int nIni = (int)roundf((float)W / (float)H);
    const float hX = (float)W / (float)nIni;
    for (int i = t; i < nIni; i += T) A.s1[i] = 0;
    __syncthreads();
    for (int k = t; k < nk; k += T) {
        int r = (int)((float)(keys[k].x & 0xFFFF) / hX);
        r = min(max(r, 0), nIni - 1);
        node_of[k] = r;
        atomicAdd(&A.s1[r], 1);
    }
    __syncthreads();
    if (t == 0) {
        int n = 0;
        for (int i = 0; i < nIni; i++) {
            if (A.s1[i] > 0) {
                const int x0 = (int)(hX * (float)i), x1 = (int)(hX * (float)(i + 1));
                A.rectA[n] = make_int2(x0, x1 | (H << 16));
                A.nA[n] = A.s1[i];
                A.s2[i] = n++;
            } else A.s2[i] = -1;
        }
        A.scan_tmp[256] = n;
    }
    __syncthreads();
    int Lsz = A.scan_tmp[256];
    __syncthreads();
    for (int k = t; k < nk; k += T) node_of[k] = A.s2[node_of[k]];
    __syncthreads();
    // ---- sweeps
    bool finish = false;
    int ncand = 0;
    int guard = 0;
    while (!finish && guard++ < 64) {
        const int prev = Lsz;
        // phase 1: divide every node with more than one key, in list order
        for (int i = t; i < Lsz; i += T) A.s1[i] = A.nA[i] > 1 ? 1 : 0;
        __syncthreads();
        const int S = plf_block_excl_scan(A.s1, Lsz, A.scan_tmp);
        for (int i = t; i < Lsz; i += T) if (A.nA[i] > 1) A.work[A.s1[i]] = i;
        __syncthreads();
        Lsz = octree_divide(A, Lsz, S, -1, keys, node_of, quad, nk, &ncand);
        if (Lsz >= N || Lsz == prev) finish = true;
        else if (Lsz + 3 * ncand > N) {
            int guard2 = 0;
            while (!finish && guard2++ < 4096) {
                const int prev2 = Lsz;
                // sort expandable nodes descending by (key count, creation rank)
                int P2 = 1;
                while (P2 < ncand) P2 <<= 1;
                for (int i = t; i < P2; i += T)
                    A.skey[i] = i < ncand ? (((unsigned long long)A.nA[A.cand[i]] << 32) | ((unsigned long long)i << 12) | 1ull) : 0ull;
                __syncthreads();
                for (int k2 = 2; k2 <= P2; k2 <<= 1)
                    for (int j = k2 >> 1; j > 0; j >>= 1) {
                        for (int i = t; i < P2; i += T) {
                            const int ixj = i ^ j;
                            if (ixj > i) {
                                const unsigned long long a = A.skey[i], b = A.skey[ixj];
                                const bool desc = (i & k2) == 0;
                                if (desc ? (a < b) : (a > b)) { A.skey[i] = b; A.skey[ixj] = a; }
                            }
                        }
                        __syncthreads();
                    }
                for (int i = t; i < ncand; i += T) A.s1[i] = A.cand[(int)((A.skey[i] >> 12) & 0xFFFFF)];
                __syncthreads();
                for (int i = t; i < ncand; i += T) A.work[i] = A.s1[i];
                __syncthreads();
                Lsz = octree_divide(A, Lsz, ncand, N, keys, node_of, quad, nk, &ncand);
                if (Lsz >= N || Lsz == prev2) finish = true;
            }
        }
    }
    // ---- keep the best key of every node (max response, first key wins ties)
    for (int i = t; i < Lsz; i += T) A.skey[i] = 0ull;
    __syncthreads();
    for (int k = t; k < nk; k += T)
        atomicMax(&A.skey[node_of[k]], ((unsigned long long)keys[k].y << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)k));
    __syncthreads();
    uint2 *so = sel + (size_t)f * g.sel_stride + L.sel_off;
    for (int i = t; i < Lsz && i < (int)L.sel_cap; i += T) {
        const unsigned k = 0xFFFFFFFFu - (unsigned)(A.skey[i] & 0xFFFFFFFFull);
        so[i] = keys[k];
    }
    if (t == 0) {
        selcnt[f * g.nlevels + l] = min(Lsz, (int)L.sel_cap);
        if (Lsz > (int)L.sel_cap) atomicOr(status, 4);
    }
}
