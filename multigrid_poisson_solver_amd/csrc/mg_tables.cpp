// mg_tables.cpp -- host-side 1-D index/weight tables for restriction and prolongation.
//
// Bit-exact indexing is the contract (BASELINE.json north_star): the reference derives
// its gather indices from fp64 floor/fmod/ceil expressions, so the tables are computed
// HERE, on the host, in fp64 with libm, with the reference's exact expressions, and the
// device kernels are pure gathers over them (hard part 3).
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "mg_internal.h"

namespace mg {

// doRestriction, src/MG_solver_CPU.cpp:647-648 (spacing 1/(N-1), 1/(M-1); L ignored)
// and :661-664 (ix_f = (int)floor(ix_c*h_c/h_f), a = fmod(ix_c*h_c, h_f)/h_f).  The
// same expressions serve rows and columns.
void build_restriction_table(int N, int M, int *lo, double *w)
{
    const double h_f = 1.0 / (double)(N - 1);
    const double h_c = 1.0 / (double)(M - 1);
    for (int i = 0; i < M; ++i) {
        lo[i] = (int)std::floor((double)i * h_c / h_f);
        w[i] = std::fmod((double)i * h_c, h_f) / h_f;
    }
}

// every interior coarse point must read inside the fine array
bool restriction_table_in_bounds(int N, int M, const int *lo)
{
    for (int i = 1; i < M - 1; ++i)
        if (lo[i] < 0 || lo[i] + 1 > N - 1) return false;
    return true;
}

// doProlongation, src/MG_solver_CPU.cpp:682-724.  N coarse, M fine, L = 1.0 (:683).
// The reference loops over coarse cells and writes the fine indices
//   [ceil(i*ratio), ceil((i+1)*ratio))                                   (:697-698)
// plus, from the cell that holds fine index M-2, the last index M-1:
//   columns (axis 1): with f_x = L, possibly overwritten by a regular write of M-1
//                     later in the same l loop                            (:701-704)
//   rows    (axis 0): with f_y = (M-1)*f_dx, after which that cell's k loop ends
//                                                                          (:706-718)
// Replaying those loops along one axis gives, per fine index, the owning cell and the
// coordinate the reference used; the kernel needs (c_hi - f) and (f - c_lo) with
// c_lo = i*c_dx, c_hi = c_lo + c_dx (:690-693).
void build_prolongation_table(int N, int M, int axis, int *owner, double *w_hi, double *w_lo)
{
    const double L = 1.0;
    const double c_dx = L / (double)(N - 1), f_dx = L / (double)(M - 1);
    const double ratio = c_dx / f_dx;
    std::vector<double> coord((size_t)M, 0.0);
    for (int k = 0; k < M; ++k) owner[k] = -1;

    for (int i = 0; i < N - 1; ++i) {
        const double end = std::ceil((i + 1) * ratio);
        for (int k = (int)std::ceil(i * ratio); k < end; ++k) {
            if (k < 0 || k >= M) continue;  // cannot happen for M >= N; keeps the table safe
            owner[k] = i;
            coord[k] = k * f_dx;
            if (k == M - 2) {
                owner[M - 1] = i;
                if (axis == 1) {
                    coord[M - 1] = L;   // later possibly overwritten by k == M-1 below
                } else {
                    coord[M - 1] = (M - 1) * f_dx;
                    break;              // the reference sets k = M-1: the loop is over
                }
            }
        }
    }
    for (int k = 0; k < M; ++k) {
        if (owner[k] < 0) {
            w_hi[k] = w_lo[k] = 0.0;
            continue;
        }
        const double c_lo = owner[k] * c_dx;
        const double c_hi = c_lo + c_dx;
        w_hi[k] = c_hi - coord[k];
        w_lo[k] = coord[k] - c_lo;
    }
}

namespace {
template <typename T>
T *upload(const std::vector<T> &v)
{
    T *d = nullptr;
    if (!MG_HIP(hipMalloc((void **)&d, v.size() * sizeof(T)))) return nullptr;
    // tables are built before first use; a synchronous copy keeps their lifetime simple
    if (!MG_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice))) return nullptr;
    return d;
}
}  // namespace

const RestrictTable &restrict_table(int N, int M)
{
    Context &c = ctx();
    auto key = std::make_pair(N, M);
    auto it = c.rtab.find(key);
    if (it != c.rtab.end()) return it->second;
    std::vector<int> lo((size_t)M);
    std::vector<double> w((size_t)M);
    build_restriction_table(N, M, lo.data(), w.data());
    RestrictTable t;
    if (!restriction_table_in_bounds(N, M, lo.data())) {
        fail(MG_ERR_ARG, "doRestriction: fine grid %d / coarse grid %d would read out of bounds", N, M);
    } else {
        t.lo = upload(lo);
        t.w = upload(w);
        // inverse map for the fused restriction: fine index -> the interior coarse index whose
        // lower-left sample it is.  Usable when consecutive samples are >= 2 apart.
        std::vector<int> inv((size_t)N, -1);
        std::vector<double> inv_w((size_t)N, 0.0);
        bool ok = M >= 3;
        for (int i = 1; i < M - 1; ++i) {
            if (i > 1 && lo[i] - lo[i - 1] < 2) ok = false;
            inv[(size_t)lo[i]] = i;
            inv_w[(size_t)lo[i]] = w[i];
        }
        t.inv = upload(inv);
        t.inv_w = upload(inv_w);
        std::vector<float> w_f(w.begin(), w.end()), inv_w_f(inv_w.begin(), inv_w.end());  // RN to fp32
        t.w_f = upload(w_f);
        t.inv_w_f = upload(inv_w_f);
        t.fusable = ok;
    }
    return c.rtab.emplace(key, t).first->second;
}

const ProlongTable &prolong_table(int N, int M)
{
    Context &c = ctx();
    auto key = std::make_pair(N, M);
    auto it = c.ptab.find(key);
    if (it != c.ptab.end()) return it->second;
    std::vector<int> orow((size_t)M), ocol((size_t)M);
    std::vector<double> rh((size_t)M), rl((size_t)M), ch((size_t)M), cl((size_t)M);
    build_prolongation_table(N, M, 0, orow.data(), rh.data(), rl.data());
    build_prolongation_table(N, M, 1, ocol.data(), ch.data(), cl.data());
    ProlongTable t;
    bool ok = true, fusable = true;
    for (int k = 0; k < M; ++k) {
        if (orow[k] > N - 2 || ocol[k] > N - 2) ok = false;
        if (orow[k] < 0 || ocol[k] < 0) fusable = false;
        if (k > 0 && (ocol[k] - ocol[k - 1] < 0 || ocol[k] - ocol[k - 1] > 1)) fusable = false;
        if (k > 0 && (orow[k] - orow[k - 1] < 0 || orow[k] - orow[k - 1] > 1)) fusable = false;
    }
    if (!ok) {
        fail(MG_ERR_ARG, "doProlongation: table for %d -> %d is out of bounds", N, M);
    } else {
        t.owner_row = upload(orow);
        t.owner_col = upload(ocol);
        t.row_hi = upload(rh);
        t.row_lo = upload(rl);
        t.col_hi = upload(ch);
        t.col_lo = upload(cl);
        std::vector<float> rhf(rh.begin(), rh.end()), rlf(rl.begin(), rl.end()), chf(ch.begin(), ch.end()), clf(cl.begin(), cl.end());
        t.row_hi_f = upload(rhf);
        t.row_lo_f = upload(rlf);
        t.col_hi_f = upload(chf);
        t.col_lo_f = upload(clf);
        t.c_dx = 1.0 / (double)(N - 1);
        t.fusable = fusable;
        t.closed_form = fusable && M <= 4096 && M >= 2;
        for (int k = 0; k < M && t.closed_form; ++k) {
            const long long q = ((long long)k * (N - 1)) / (M - 1);
            const int want = (int)(q < N - 2 ? q : N - 2);
            if (orow[k] != want || ocol[k] != want) t.closed_form = false;
        }
        // four consecutive fine columns starting on a multiple of 4 span at most three coarse cells: what the
        // 4-columns-per-lane form of the fused prolongation (fp32 fields) holds per lane
        t.fusable4 = fusable && M % 4 == 0;
        for (int k = 0; k + 3 < M && t.fusable4; k += 4)
            if (ocol[k + 3] - ocol[k] > 2) t.fusable4 = false;
    }
    return c.ptab.emplace(key, t).first->second;
}

void parallel_for(size_t n, void (*fn)(size_t, size_t, void *), void *arg, size_t serial_below)
{
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? hw : 4;
    if (nt > 64) nt = 64;
    if (n < serial_below || nt == 1) {
        fn(0, n, arg);
        return;
    }
    std::vector<std::thread> th;
    const size_t chunk = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        const size_t b = t * chunk, e = b + chunk < n ? b + chunk : n;
        if (b >= e) break;
        th.emplace_back(fn, b, e, arg);
    }
    for (auto &t : th) t.join();
}

}  // namespace mg
