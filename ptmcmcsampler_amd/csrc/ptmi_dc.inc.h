// ptmi_dc.inc.h -- eigenvalues and eigenvectors of a symmetric TRIDIAGONAL matrix by divide and conquer, and the back-transformation
// through the Householder reflectors of sytrd_lds_kernel: the second half of ptmi_eig_sytrd (PTMCMCSampler.py:797-803 calls LAPACK's
// SVD once per covariance epoch; round 4 handed the tridiagonal matrix to rocsolver_dstedc / rocsolver_dormtr).
//
// Cuppen's scheme with the Gu-Eisenstat stabilisation (the method of LAPACK's dstedc, restated for the device):
//   * the matrix is split in halves down to leaves of <= 16 rows (T = diag(T1', T2') + |rho| v v^T at every split, rho the
//     off-diagonal entry cut); a leaf is solved by implicit QL in one wave;
//   * a merge sorts the children's eigenvalues, DEFLATES (a tiny component of z = the adjoining rows of the children's eigenvectors,
//     or two close eigenvalues after a plane rotation: LAPACK dlaed2's test) -- the nearly degenerate spectra an isotropic target
//     adapts to deflate almost completely -- and finds the remaining k eigenvalues as roots of the secular equation
//     1 + rho sum_i z_i^2 / (d_i - lambda) = 0 by BISECTION in coordinates shifted to the closer pole (a wave per root, the sum over
//     the lanes): monotone on its interval, no safeguards to get wrong, and the differences d_i - lambda_j come out to full relative
//     accuracy, which is what the stabilisation needs;
//   * z is then RECOMPUTED from the roots (Loewner's formula), so that the eigenvectors u_j = (z^_i / (d_i - lambda_j))_i of the
//     rank-one update are orthogonal to working precision however close the roots are, and the children's vectors are multiplied
//     by them on the matrix cores (v_mfma_f64_16x16x4_f64).
// Checked against numpy on random, Wilkinson, nearly degenerate and near-identity matrices (tests/test_eig_jacobi.py): eigenvalues,
// orthogonality and residual at a few 1e-15.  Its last bits are its own (no oracle restates it), like the library's before it.
#pragma once

namespace dc {

#ifndef PTMI_DC_LEAF
#define PTMI_DC_LEAF 16
#endif
constexpr int LEAF = PTMI_DC_LEAF;       // rows of a leaf (<= 32: a leaf is solved by one wave, lane = row)
constexpr int NMAX = 1024;

struct Node { int off, n, n1; };         // a merge: children [off, off + n1) and [off + n1, off + n)
struct Leaf { int off, n; };

// everything a level's kernels share; arrays of n doubles / ints are indexed by the global row (off + local), so the nodes of a
// level never overlap
struct Args {
    int n;
    const Node *nodes;                   // the level's merges
    double *d, *e;                       // the tridiagonal matrix (d modified at the splits)
    double *Din, *Dout;                  // eigenvalues of the children / of the merged nodes
    double *Qin, *Qout;                  // eigenvectors, vector-major: vector v of the matrix is Q[v * n + 0 .. n)
    double *U;                           // eigenvectors of the rank-one updates, U[(off + j) * n + i]: component i of vector j
    double *dk, *zk, *Ddefl;             // kept poles and weights (sorted), deflated eigenvalues
    int *keepv, *deflv;                  // ... the local vector each came from
    int *cnt;                            // [2 * node]: k, number deflated
    double *rho;                         // [node]: 2 |rho| (z normalised)
    int *org;                            // the pole a root is measured from
    double *mu, *lam, *zh;               // root - pole, root, recomputed weights
    int *rankk, *rankd;                  // position of a root / of a deflated eigenvalue in the merged node's ascending order
};

// the rank-one modifications of all splits: d[cut - 1] -= |e[cut - 1]|, d[cut] -= |e[cut - 1]|
__global__ void split_kernel(const Node *nodes, int nnodes, double *d, const double *e)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= nnodes) return;
    const int cut = nodes[t].off + nodes[t].n1;
    const double r = __builtin_fabs(e[cut - 1]);
    d[cut - 1] -= r;
    d[cut] -= r;
}

__device__ __forceinline__ double hyp(double a, double b)     // sqrt(a^2 + b^2) without overflow
{
    const double x = __builtin_fabs(a), y = __builtin_fabs(b);
    const double hi = x > y ? x : y, lo = x > y ? y : x;
    if (hi == 0.0) return 0.0;
    const double q = lo / hi;
    return hi * det_sqrt(1.0 + q * q);
}

// a leaf: implicit QL with eigenvectors (the classical tql2 / tqli recurrence) in one wave, the vectors in LDS (lane r owns row r)
// info: set to (leaf + 1) when a leaf's QL iteration has not converged within 60 sweeps of an eigenvalue (LAPACK's dsteqr gives up at 30 n
// in all; the merges cannot fail: their roots come from a bisection that ends when the interval cannot shrink)
__global__ __launch_bounds__(64) void leaf_kernel(const Leaf *leaves, int n, const double *d, const double *e, double *Dout, double *Qout, int *info)
{
    __shared__ double z[LEAF][LEAF + 1], dd[LEAF], ee[LEAF];
    const Leaf lf = leaves[blockIdx.x];
    const int m0 = lf.n, off = lf.off, r = (int)threadIdx.x;
    if (r < m0) {
        dd[r] = d[off + r];
        ee[r] = r + 1 < m0 ? e[off + r] : 0.0;
        for (int c = 0; c < m0; ++c) z[r][c] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    bool failed = false;
    for (int l = 0; l < m0; ++l) {
        bool conv = false;
        for (int iter = 0; iter <= 60; ++iter) {
            int m = l;
            for (; m < m0 - 1; ++m) {
                const double s = __builtin_fabs(dd[m]) + __builtin_fabs(dd[m + 1]);
                if (__builtin_fabs(ee[m]) <= 2.220446049250313e-16 * s) break;
            }
            if (m == l) { conv = true; break; }
            if (iter == 60) break;                      // 60 sweeps done and the eigenvalue has not split off
            double g = (dd[l + 1] - dd[l]) / (2.0 * ee[l]);
            double rr = hyp(g, 1.0);
            g = dd[m] - dd[l] + ee[l] / (g + (g >= 0.0 ? rr : -rr));
            double s = 1.0, c = 1.0, p = 0.0;
            int i = m - 1;
            bool under = false;
            for (; i >= l; --i) {
                double f = s * ee[i];
                const double b = c * ee[i];
                rr = hyp(f, g);
                __syncthreads();
                if (r == 0) ee[i + 1] = rr;
                if (rr == 0.0) {
                    if (r == 0) { dd[i + 1] -= p; ee[m] = 0.0; }
                    under = true;
                    break;
                }
                s = f / rr;
                c = g / rr;
                g = dd[i + 1] - p;
                rr = (dd[i] - g) * s + 2.0 * c * b;
                p = s * rr;
                __syncthreads();
                if (r == 0) dd[i + 1] = g + p;
                g = c * rr - b;
                if (r < m0) {
                    f = z[r][i + 1];
                    z[r][i + 1] = s * z[r][i] + c * f;
                    z[r][i] = c * z[r][i] - s * f;
                }
            }
            __syncthreads();
            if (under) continue;
            if (r == 0) { dd[l] -= p; ee[l] = g; ee[m] = 0.0; }
            __syncthreads();
        }
        failed |= !conv;
    }
    if (failed && r == 0 && info != nullptr) atomicMax(info, (int)blockIdx.x + 1);
    __syncthreads();
    if (r < m0) {
        Dout[off + r] = dd[r];
        for (int c = 0; c < m0; ++c) Qout[(size_t)(off + c) * n + off + r] = z[r][c];      // vector c, component r
    }
}

// ---- a merge, step 1 (one block per node): z, the sort, the deflation scan, the plane rotations of deflated pairs
constexpr int PREP_THREADS = 1024;
__global__ __launch_bounds__(PREP_THREADS) void prep_kernel(Args a)
{
    __shared__ double Ds[NMAX], zs[NMAX], rc[NMAX], rs[NMAX];
    __shared__ int perm[NMAX], rp[NMAX], rj[NMAX], keep[NMAX], defl[NMAX];
    __shared__ int sh_k, sh_nd, sh_nrot;
    __shared__ double sh_rho;
    const Node nd = a.nodes[blockIdx.x];
    const int off = nd.off, nn = nd.n, n1 = nd.n1, n = a.n, t = (int)threadIdx.x;
    const double e_cut = a.e[off + n1 - 1];
    const double sgn = e_cut < 0.0 ? -1.0 : 1.0;
    // the children's adjoining rows: the last component of child 1's vectors, the first of child 2's
    double Dt = 0.0, zt = 0.0;
    if (t < nn) {
        Dt = a.Din[off + t];
        zt = t < n1 ? a.Qin[(size_t)(off + t) * n + off + n1 - 1] : sgn * a.Qin[(size_t)(off + t) * n + off + n1];
        zt *= 0.70710678118654752440;                                  // |z| = sqrt 2 -> 1 (rho doubles)
        Ds[t] = Dt;                                                    // staged unsorted first
    }
    __syncthreads();
    int rank = 0;
    if (t < nn) {
        for (int j = 0; j < nn; ++j) { const double o = Ds[j]; rank += (o < Dt) || (o == Dt && j < t); }
    }
    __syncthreads();
    if (t < nn) { Ds[rank] = Dt; zs[rank] = zt; perm[rank] = t; }
    __syncthreads();
    if (t == 0) {
        const double rho = 2.0 * __builtin_fabs(e_cut);
        double dmax = 0.0, zmax = 0.0;
        for (int j = 0; j < nn; ++j) {
            dmax = __builtin_fabs(Ds[j]) > dmax ? __builtin_fabs(Ds[j]) : dmax;
            zmax = __builtin_fabs(zs[j]) > zmax ? __builtin_fabs(zs[j]) : zmax;
        }
        const double tol = 8.0 * 2.220446049250313e-16 * (dmax > zmax ? dmax : zmax);
        int k = 0, ndf = 0, nrot = 0, pj = -1;
        if (rho * zmax <= tol) {
            for (int j = 0; j < nn; ++j) defl[ndf++] = j;              // nothing couples the halves
        } else {
            for (int j = 0; j < nn; ++j) {
                if (rho * __builtin_fabs(zs[j]) <= tol) { defl[ndf++] = j; continue; }
                if (pj < 0) { pj = j; continue; }
                double s = zs[pj], c = zs[j];
                const double tau = hyp(c, s), dt = Ds[j] - Ds[pj];
                c /= tau;
                s = -s / tau;
                if (__builtin_fabs(dt * c * s) <= tol) {               // close pair: rotate z[pj] away (dlaed2)
                    zs[j] = tau;
                    zs[pj] = 0.0;
                    rp[nrot] = pj; rj[nrot] = j; rc[nrot] = c; rs[nrot] = s;
                    ++nrot;
                    const double dp = Ds[pj], dj = Ds[j];
                    Ds[pj] = dp * c * c + dj * s * s;
                    Ds[j] = dp * s * s + dj * c * c;
                    defl[ndf++] = pj;
                    pj = j;
                } else {
                    keep[k++] = pj;
                    pj = j;
                }
            }
            if (pj >= 0) keep[k++] = pj;
        }
        sh_k = k; sh_nd = ndf; sh_nrot = nrot; sh_rho = rho;
    }
    __syncthreads();
    // the rotations on the vectors: thread = component; a chain (pj, j), (j, j'), ... carries its second vector in a register
    const int nrot = sh_nrot;
    for (int r = t; r < nn; r += PREP_THREADS) {
        int have = -1;
        double carry = 0.0;
        for (int q = 0; q < nrot; ++q) {
            const int pj = rp[q], j = rj[q];
            double *vp = a.Qin + (size_t)(off + perm[pj]) * n + off + r, *vj = a.Qin + (size_t)(off + perm[j]) * n + off + r;
            const double qp = have == pj ? carry : *vp, qj = *vj;
            const double c = rc[q], s = rs[q];
            *vp = c * qp + s * qj;
            carry = -s * qp + c * qj;
            have = j;
            if (q + 1 == nrot || rp[q + 1] != j) { *vj = carry; have = -1; }
        }
    }
    const int k = sh_k, ndf = sh_nd;
    for (int i = t; i < k; i += PREP_THREADS) {
        a.dk[off + i] = Ds[keep[i]];
        a.zk[off + i] = zs[keep[i]];
        a.keepv[off + i] = perm[keep[i]];
    }
    for (int i = t; i < ndf; i += PREP_THREADS) {
        a.Ddefl[off + i] = Ds[defl[i]];
        a.deflv[off + i] = perm[defl[i]];
    }
    if (t == 0) { a.cnt[2 * blockIdx.x] = k; a.cnt[2 * blockIdx.x + 1] = ndf; a.rho[blockIdx.x] = sh_rho; }
}

__device__ __forceinline__ double wave_sum(double p)
{
    p = sum_xor32(p);
    p = sum_xor16(p);
    p = p + dppf64<0x128>(p);
    p = p + dppf64<0x124>(p);
    p = p + dppf64<0x4E>(p);
    p = p + dppf64<0xB1>(p);
    return p;
}
__device__ __forceinline__ double wave_prod(double p)
{
    p = p * __shfl_xor(p, 32, 64);
    p = p * __shfl_xor(p, 16, 64);
    p = p * __shfl_xor(p, 8, 64);
    p = p * __shfl_xor(p, 4, 64);
    p = p * __shfl_xor(p, 2, 64);
    p = p * __shfl_xor(p, 1, 64);
    return p;
}

// ---- step 2: the secular roots.  grid (ceil(nmax / 4), nodes), a wave per root: root j lies in (d_j, d_j+1) (the last one in
// (d_k-1, d_k-1 + rho |z|^2)); measured from the closer pole it is found by bisection on f(mu) = 1 + rho sum_i z_i^2 / (delta_i - mu)
constexpr int PT = NMAX / 64;            // poles per lane
__global__ __launch_bounds__(256) void secular_kernel(Args a)
{
    const Node nd = a.nodes[blockIdx.y];
    const int off = nd.off, k = a.cnt[2 * blockIdx.y], lane = (int)(threadIdx.x & 63);
    const int j = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= k) return;
    const double rho = a.rho[blockIdx.y];
    double dl[PT], zz[PT];
    double zsum = 0.0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int i = lane + 64 * u;
        dl[u] = i < k ? a.dk[off + i] : 0.0;
        const double z = i < k ? a.zk[off + i] : 0.0;
        zz[u] = z * z;
        zsum += zz[u];
    }
    zsum = wave_sum(zsum);
    const double dj = a.dk[off + j];
    auto f_at = [&](double o, double m) {                             // 1 + rho sum zz_i / ((d_i - o) - m)
        double p = 0.0;
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = lane + 64 * u;
            const double den = (dl[u] - o) - m;
            p += i < k ? zz[u] / den : 0.0;
        }
        return 1.0 + rho * wave_sum(p);
    };
    int o_idx = j;
    double lo, hi, o = dj;
    if (j < k - 1) {
        const double dn = a.dk[off + j + 1], mid = 0.5 * (dn - dj);
        if (f_at(dj, mid) > 0.0) { lo = 0.0; hi = mid; }              // the root is nearer to d_j
        else { o_idx = j + 1; o = dn; lo = -mid; hi = 0.0; }
    } else {
        lo = 0.0;
        hi = rho * zsum;
    }
    for (int it = 0; it < 1200; ++it) {
        const double m = 0.5 * (lo + hi);
        if (m == lo || m == hi) break;
        if (f_at(o, m) > 0.0) hi = m;
        else lo = m;
    }
    if (lane == 0) {
        const double mu = 0.5 * (lo + hi);
        a.org[off + j] = o_idx;
        a.mu[off + j] = mu;
        a.lam[off + j] = o + mu;
    }
}

// d_i - lambda_j from the shifted root: (d_i - d_org(j)) - mu_j
__device__ __forceinline__ double dminus(const Args &a, int off, int i, int j) { return (a.dk[off + i] - a.dk[off + a.org[off + j]]) - a.mu[off + j]; }

// ---- step 3: z^_i = sign(z_i) sqrt(prod_j (lambda_j - d_i) / prod_{j != i} (d_j - d_i) / rho) (a wave per i)
__global__ __launch_bounds__(256) void zhat_kernel(Args a)
{
    const Node nd = a.nodes[blockIdx.y];
    const int off = nd.off, k = a.cnt[2 * blockIdx.y], lane = (int)(threadIdx.x & 63);
    const int i = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (i >= k) return;
    const double di = a.dk[off + i];
    double p = 1.0;
    for (int j = lane; j < k; j += 64) {
        const double num = -dminus(a, off, i, j);                      // lambda_j - d_i
        p *= j == i ? num : num / (a.dk[off + j] - di);
    }
    p = wave_prod(p);
    if (lane == 0) {
        const double v = det_sqrt(__builtin_fabs(p) / a.rho[blockIdx.y]);
        a.zh[off + i] = a.zk[off + i] < 0.0 ? -v : v;
    }
}

// ---- step 4: the vectors of the rank-one update, normalised (a wave per vector j): U[(off + j) n + i] = z^_i / (d_i - lambda_j)
__global__ __launch_bounds__(256) void vectors_kernel(Args a)
{
    const Node nd = a.nodes[blockIdx.y];
    const int off = nd.off, k = a.cnt[2 * blockIdx.y], lane = (int)(threadIdx.x & 63), n = a.n;
    const int j = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (j >= k) return;
    double v[PT], ss = 0.0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int i = lane + 64 * u;
        v[u] = i < k ? a.zh[off + i] / dminus(a, off, i, j) : 0.0;
        ss += v[u] * v[u];
    }
    const double inv = 1.0 / det_sqrt(wave_sum(ss));
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int i = lane + 64 * u;
        if (i < k) a.U[(size_t)(off + j) * n + i] = v[u] * inv;
    }
}

// ---- step 5: where every eigenvalue of the merged node goes (ascending): roots and deflated values ranked by counting
__global__ __launch_bounds__(1024) void rank_kernel(Args a)
{
    __shared__ double val[NMAX];
    const Node nd = a.nodes[blockIdx.x];
    const int off = nd.off, nn = nd.n, k = a.cnt[2 * blockIdx.x], t = (int)threadIdx.x;
    double mine = 0.0;
    if (t < nn) {
        mine = t < k ? a.lam[off + t] : a.Ddefl[off + t - k];
        val[t] = mine;
    }
    __syncthreads();
    if (t < nn) {
        int rank = 0;
        for (int j = 0; j < nn; ++j) { const double o = val[j]; rank += (o < mine) || (o == mine && j < t); }
        a.Dout[off + rank] = mine;
        if (t < k) a.rankk[off + t] = rank;
        else a.rankd[off + t - k] = rank;
    }
}

// ---- step 6: the merged vectors.  Qout[rank(j)][off + r] = sum_i U[j][i] Qin[keepv(i)][off + r] on v_mfma_f64_16x16x4_f64: a block
// of four waves computes 64 vectors x 64 components, wave w the vectors 16 w .. 16 w + 15 (the A fragment of a k-step serves its four
// tiles); operands through LDS in chunks of 16 poles.  grid (ceil(nmax / 64), ceil(nmax / 64), nodes).
typedef double dc_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gemm_kernel(Args a)
{
    __shared__ double As[64][17], Bs[16][65];
    const Node nd = a.nodes[blockIdx.z];
    const int off = nd.off, nn = nd.n, k = a.cnt[2 * blockIdx.z], n = a.n;
    const int j0 = (int)blockIdx.y * 64, r0 = (int)blockIdx.x * 64;
    if (j0 >= k || r0 >= nn) return;
    const int t = (int)threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, l4 = lane >> 4;
    dc_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = dc_d4{0.0, 0.0, 0.0, 0.0};
    for (int i0 = 0; i0 < k; i0 += 16) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                  // A chunk: 64 vectors x 16 poles
            const int idx = t + 256 * u, jj = idx >> 4, ii = idx & 15;
            As[jj][ii] = (j0 + jj < k && i0 + ii < k) ? a.U[(size_t)(off + j0 + jj) * n + i0 + ii] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                  // B chunk: 16 poles x 64 components
            const int idx = t + 256 * u, ii = idx >> 6, rr = idx & 63;
            double v = 0.0;
            if (i0 + ii < k && r0 + rr < nn) v = a.Qin[(size_t)(off + a.keepv[off + i0 + ii]) * n + off + r0 + rr];
            Bs[ii][rr] = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const double av = As[16 * w + l15][4 * ks + l4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[4 * ks + l4][16 * q + l15], acc[q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = j0 + 16 * w + l4 + 4 * reg, r = r0 + 16 * q + l15;
            if (j < k && r < nn) a.Qout[(size_t)(off + a.rankk[off + j]) * n + off + r] = acc[q][reg];
        }
}

// the deflated vectors move to their places unchanged
__global__ __launch_bounds__(256) void copy_deflated_kernel(Args a)
{
    const Node nd = a.nodes[blockIdx.y];
    const int off = nd.off, nn = nd.n, ndf = a.cnt[2 * blockIdx.y + 1], n = a.n;
    const int v = (int)blockIdx.x;
    if (v >= ndf) return;
    const double *src = a.Qin + (size_t)(off + a.deflv[off + v]) * n + off;
    double *dst = a.Qout + (size_t)(off + a.rankd[off + v]) * n + off;
    for (int r = (int)threadIdx.x; r < nn; r += 256) dst[r] = src[r];
}

// ---- the back-transformation: Z := Q Z with Q = H(0) H(1) ... H(n-2) of the tridiagonalization (LAPACK's dsytrd storage, lower:
// reflector m is v = (0 .. 0, 1 at m + 1, A[m][m + 2 .. n)) with tau[m]).  Every eigenvector is transformed on its own: a wave holds
// VPW vectors in registers and applies H(n-2) ... H(0) to them in turn; the reflectors stream from L2.
#ifndef PTMI_DC_VPW
#define PTMI_DC_VPW 1
#endif
constexpr int VPW = PTMI_DC_VPW;        // vectors per wave: 1000 independent dependent chains want waves, not registers (1: 0.x ms; 4: 2.0 ms on 63 CUs)
__global__ __launch_bounds__(256) void backtransform_kernel(const double *A, const double *tau, int n, double *Z)
{
    const int lane = (int)(threadIdx.x & 63);
    const int v0 = (int)((blockIdx.x * 4 + (threadIdx.x >> 6)) * VPW);
    if (v0 >= n) return;
    double z[VPW][PT];
#pragma unroll
    for (int q = 0; q < VPW; ++q)
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = lane + 64 * u;
            z[q][u] = (v0 + q < n && i < n) ? Z[(size_t)(v0 + q) * n + i] : 0.0;
        }
    // reflector m - 1 is requested while reflector m is applied (a row comes from L2: its round trip was the step's critical path).
    // The loads are unconditional (clamped addresses; what lies below the reflector's first element is selected away): a load under
    // `i >= m + 2` was a branch of its own for every slot
    int ic[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) ic[u] = lane + 64 * u < n ? lane + 64 * u : n - 1;
    auto fetch = [&](int m, double (&raw)[PT]) {
        const double *row = A + (size_t)(m < 0 ? 0 : m) * n;
#pragma unroll
        for (int u = 0; u < PT; ++u) raw[u] = row[ic[u]];
    };
    auto apply = [&](int m, const double (&raw)[PT]) {
        const double tm = tau[m];
        double vv[PT];
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = lane + 64 * u;
            vv[u] = (i >= m + 2 && i < n) ? raw[u] : (i == m + 1 ? 1.0 : 0.0);
        }
#pragma unroll
        for (int q = 0; q < VPW; ++q) {
            double p0 = 0.0, p1 = 0.0;                                 // two chains: the sixteen products of a lane are no single dependent chain
#pragma unroll
            for (int u = 0; u < PT; u += 2) { p0 = __builtin_fma(vv[u], z[q][u], p0); p1 = __builtin_fma(vv[u + 1], z[q][u + 1], p1); }
            const double s = tm * wave_sum(p0 + p1);
#pragma unroll
            for (int u = 0; u < PT; ++u) z[q][u] = __builtin_fma(-s, vv[u], z[q][u]);
        }
    };
    double va[PT], vb[PT];
    fetch(n - 2, va);
    for (int m = n - 2; m >= 0; m -= 2) {
        fetch(m - 1, vb);
        apply(m, va);
        if (m - 1 < 0) break;
        fetch(m - 2, va);
        apply(m - 1, vb);
    }
#pragma unroll
    for (int q = 0; q < VPW; ++q)
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = lane + 64 * u;
            if (v0 + q < n && i < n) Z[(size_t)(v0 + q) * n + i] = z[q][u];
        }
}

}  // namespace dc
