// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product.
//
// Single-threaded CPU restatement of ztlpn/minilp 0.2.2's simplex hot path
// (src/sparse.rs, src/ordering.rs:4-21,393-460, src/lu.rs, src/solver.rs,
// the Problem/Solution layer of src/lib.rs and the parser of src/mps.rs).
// Every function cites the reference file:line whose behaviour it restates:
// iteration order, strict comparisons, tie-breaks, EPS placement and the
// eta-file zero padding are kept exactly.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library. The product (minilp_amd/) never links or calls it.
//
// Parity pin: the Rust reference cannot be built in this image (no cargo,
// sprs not vendored), so this restatement is pinned by the reference's own
// known-answer tests (tests/test_oracle_kat.py reproduces every assert of
// lib.rs:470-645, solver.rs:1391-1479, lu.rs:479-704, sparse.rs:344-359,
// mps.rs:437-476) and cross-checked against HiGHS objective fixtures.
//
// Third-party boundary: sprs 0.9.2 (Cargo.toml:14) is storage only. The
// behaviours relied on are restated in CsVec/CsMat below: CsVec::new sorts
// (index,value) pairs and rejects duplicates; to_csc() yields ascending row
// indices per column; squared_l2_norm is a left-to-right sum of squares.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <time.h>

typedef size_t usize;
static const double EPS = 1e-8;  // solver.rs:12
static const double INF = std::numeric_limits<double>::infinity();

struct OraclePanic : std::runtime_error {
    explicit OraclePanic(const std::string& s) : std::runtime_error(s) {}
};
enum LpError { LP_OK = 0, LP_INFEASIBLE = 1, LP_UNBOUNDED = 2 };  // lib.rs:172-178
struct LpFail { LpError e; };
struct SingularMatrix {};  // sparse.rs:335-338

// ---------------------------------------------------------------- sprs bits
// sprs::CsVecI<f64,usize>: sorted sparse vector (used at lib.rs:279, 376).
struct CsVec {
    usize dim = 0;
    std::vector<usize> indices;
    std::vector<double> data;
    // sprs CsVec::new: sorts pairs by index, panics on duplicates / out of range.
    static CsVec make(usize dim, std::vector<usize> idx, std::vector<double> val) {
        if (idx.size() != val.size()) throw OraclePanic("CsVec::new: length mismatch");
        std::vector<usize> ord(idx.size());
        for (usize i = 0; i < ord.size(); ++i) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](usize a, usize b) { return idx[a] < idx[b]; });
        CsVec v;
        v.dim = dim;
        v.indices.resize(idx.size());
        v.data.resize(idx.size());
        for (usize i = 0; i < ord.size(); ++i) {
            v.indices[i] = idx[ord[i]];
            v.data[i] = val[ord[i]];
        }
        for (usize i = 0; i + 1 < v.indices.size(); ++i)
            if (v.indices[i] == v.indices[i + 1]) throw OraclePanic("CsVec::new: duplicate index");
        if (!v.indices.empty() && v.indices.back() >= dim) throw OraclePanic("CsVec::new: index out of bounds");
        return v;
    }
    void append(usize i, double v) {  // sprs CsVec::append: increasing index
        if (!indices.empty() && i <= indices.back()) throw OraclePanic("CsVec::append: not increasing");
        if (i >= dim) throw OraclePanic("CsVec::append: out of bounds");
        indices.push_back(i);
        data.push_back(v);
    }
};

// solver.rs:1370-1384 into_resized: trim trailing indices >= len, re-dimension.
static CsVec into_resized(CsVec vec, usize len) {
    while (!vec.indices.empty()) {
        if (vec.indices.back() < len) break;
        vec.indices.pop_back();
        vec.data.pop_back();
    }
    return CsVec::make(len, vec.indices, vec.data);
}

// sprs::CsMatI<f64,usize>: compressed matrix, outer dimension grows by append.
struct CsMat {
    usize inner_dim = 0;  // number of columns for CSR, rows for CSC
    std::vector<usize> indptr{0};
    std::vector<usize> indices;
    std::vector<double> data;
    usize outer() const { return indptr.size() - 1; }
    usize nnz() const { return data.size(); }
    void append_outer(const usize* idx, const double* val, usize k) {
        indices.insert(indices.end(), idx, idx + k);
        data.insert(data.end(), val, val + k);
        indptr.push_back(indices.size());
    }
    // sprs to_csc()/to_csr(): counting transpose; ascending outer index inside each inner line.
    CsMat to_other_storage() const {
        CsMat out;
        out.inner_dim = outer();
        out.indptr.assign(inner_dim + 1, 0);
        for (usize i : indices) out.indptr[i + 1] += 1;
        for (usize i = 0; i < inner_dim; ++i) out.indptr[i + 1] += out.indptr[i];
        out.indices.resize(nnz());
        out.data.resize(nnz());
        std::vector<usize> next(out.indptr.begin(), out.indptr.end() - 1);
        for (usize o = 0; o < outer(); ++o)
            for (usize p = indptr[o]; p < indptr[o + 1]; ++p) {
                usize dst = next[indices[p]]++;
                out.indices[dst] = o;
                out.data[dst] = data[p];
            }
        return out;
    }
};

// ---------------------------------------------------------------- sparse.rs
struct SparseVec {  // sparse.rs:4-39
    std::vector<usize> indices;
    std::vector<double> values;
    void clear() { indices.clear(); values.clear(); }
    void push(usize i, double v) { indices.push_back(i); values.push_back(v); }
    usize len() const { return indices.size(); }
    double sq_norm() const {  // sparse.rs:32-34
        double s = 0.0;
        for (double v : values) s += v * v;
        return s;
    }
};

struct ScatteredVec {  // sparse.rs:41-136
    std::vector<double> values;
    std::vector<uint8_t> is_nonzero;
    std::vector<usize> nonzero;
    static ScatteredVec empty(usize n) {
        ScatteredVec s;
        s.values.assign(n, 0.0);
        s.is_nonzero.assign(n, 0);
        return s;
    }
    usize len() const { return values.size(); }
    double get(usize i) const { return values[i]; }  // sparse.rs:70-72
    double& get_mut(usize i) {                        // sparse.rs:75-80
        if (!is_nonzero[i]) {
            is_nonzero[i] = 1;
            nonzero.push_back(i);
        }
        return values[i];
    }
    double sq_norm() const {  // sparse.rs:82-87
        double s = 0.0;
        for (usize i : nonzero) s += values[i] * values[i];
        return s;
    }
    void clear() {  // sparse.rs:89-95
        for (usize i : nonzero) {
            values[i] = 0.0;
            is_nonzero[i] = 0;
        }
        nonzero.clear();
    }
    void clear_and_resize(usize n) {  // sparse.rs:97-101
        clear();
        values.resize(n, 0.0);
        is_nonzero.resize(n, 0);
    }
    void set(const usize* idx, const double* val, usize k) {  // sparse.rs:103-113
        clear();
        for (usize p = 0; p < k; ++p) {
            is_nonzero[idx[p]] = 1;
            nonzero.push_back(idx[p]);
            values[idx[p]] = val[p];
        }
    }
    void to_sparse_vec(SparseVec& lhs) const {  // sparse.rs:115-121
        lhs.clear();
        for (usize i : nonzero) lhs.push(i, values[i]);
    }
};

struct SparseMat {  // sparse.rs:138-270 (column-compressed, unordered rows)
    usize n_rows = 0;
    std::vector<usize> indptr{0};
    std::vector<usize> indices;
    std::vector<double> data;
    explicit SparseMat(usize n = 0) : n_rows(n) {}
    usize rows() const { return n_rows; }
    usize cols() const { return indptr.size() - 1; }
    usize nnz() const { return data.size(); }
    void clear_and_resize(usize n) {  // sparse.rs:169-175
        data.clear();
        indices.clear();
        indptr.assign(1, 0);
        n_rows = n;
    }
    void push(usize r, double v) { indices.push_back(r); data.push_back(v); }
    void seal_column() { indptr.push_back(indices.size()); }
    SparseMat transpose() const {  // sparse.rs:230-269: rows filled back to front
        SparseMat out(cols());
        out.indptr.assign(rows() + 1, 0);
        for (usize c = 0; c < cols(); ++c)
            for (usize p = indptr[c]; p < indptr[c + 1]; ++p) out.indptr[indices[p]] += 1;
        for (usize r = 1; r < out.indptr.size(); ++r) out.indptr[r] += out.indptr[r - 1];
        out.indices.assign(nnz(), 0);
        out.data.assign(nnz(), 0.0);
        for (usize c = 0; c < cols(); ++c)
            for (usize p = indptr[c]; p < indptr[c + 1]; ++p) {
                usize r = indices[p];
                out.indptr[r] -= 1;
                out.indices[out.indptr[r]] = c;
                out.data[out.indptr[r]] = data[p];
            }
        out.indptr.back() = nnz();
        return out;
    }
};

struct TriangleMat {  // sparse.rs:272-316
    SparseMat nondiag;
    bool has_diag = false;  // diag: None means all 1's
    std::vector<double> diag;
    usize rows() const { return nondiag.rows(); }
    usize cols() const { return nondiag.cols(); }
    TriangleMat transpose() const {
        TriangleMat t;
        t.nondiag = nondiag.transpose();
        t.has_diag = has_diag;
        t.diag = diag;
        return t;
    }
};

struct Perm {  // sparse.rs:329-333
    std::vector<usize> orig2new, new2orig;
};

// -------------------------------------------------------------- ordering.rs
struct ColsQueue {  // ordering.rs:393-460: bucket queue, FIFO inside a score
    static const usize NONE = (usize)-1;
    std::vector<usize> score2head, prev, next;
    usize min_score, len = 0;
    explicit ColsQueue(usize n) : score2head(n, NONE), prev(n, 0), next(n, 0), min_score(n) {}
    void add(usize col, usize score) {  // ordering.rs:428-443
        if (score >= score2head.size()) throw OraclePanic("ColsQueue::add: score out of range (empty column?)");
        min_score = std::min(min_score, score);
        len += 1;
        usize head = score2head[score];
        if (head != NONE) {
            prev[col] = prev[head];
            next[col] = head;
            next[prev[head]] = col;
            prev[head] = col;
        } else {
            prev[col] = col;
            next[col] = col;
            score2head[score] = col;
        }
    }
    void remove(usize col, usize score) {  // ordering.rs:445-457
        len -= 1;
        if (next[col] == col) {
            score2head[score] = NONE;
        } else {
            next[prev[col]] = next[col];
            prev[next[col]] = prev[col];
            if (score2head[score] == col) score2head[score] = next[col];
        }
    }
    bool pop_min(usize& out) {  // ordering.rs:413-426
        usize col;
        for (;;) {
            if (min_score >= score2head.size()) return false;
            if (score2head[min_score] != NONE) {
                col = score2head[min_score];
                break;
            }
            min_score += 1;
        }
        remove(col, min_score);
        out = col;
        return true;
    }
};

struct ColView {
    const usize* rows;
    const double* vals;
    usize len;
};
typedef std::function<ColView(usize)> GetCol;

static Perm order_simple(usize size, const GetCol& get_col) {  // ordering.rs:4-21
    ColsQueue q(size);
    for (usize c = 0; c < size; ++c) q.add(c, get_col(c).len - 1);  // usize underflow on empty col
    Perm p;
    p.new2orig.reserve(size);
    while (p.new2orig.size() < size) {
        usize c;
        if (!q.pop_min(c)) throw OraclePanic("order_simple: queue empty");
        p.new2orig.push_back(c);
    }
    p.orig2new.assign(size, 0);
    for (usize n = 0; n < size; ++n) p.orig2new[p.new2orig[n]] = n;
    return p;
}

// -------------------------------------------------------------------- lu.rs
struct MarkNonzero {  // lu.rs:306-407
    struct DfsStep { usize orig_i, cur_child; };
    std::vector<DfsStep> dfs_stack;
    std::vector<uint8_t> is_visited;
    std::vector<usize> visited;  // reverse topological order
    void clear() {               // lu.rs:328-334
        for (usize i : visited) is_visited[i] = 0;
        visited.clear();
    }
    void clear_and_resize(usize n) {  // lu.rs:336-340
        clear();
        is_visited.resize(n, 0);
    }
    // lu.rs:343-406
    template <class Children, class Filter, class O2N>
    void run(ScatteredVec& rhs, Children get_children, Filter filter, O2N orig2new_row) {
        clear();
        for (usize nz = 0; nz < rhs.nonzero.size(); ++nz) {
            usize orig_r = rhs.nonzero[nz];
            usize new_r = orig2new_row(orig_r);
            if (!filter(new_r)) continue;
            if (is_visited[orig_r]) continue;
            dfs_stack.push_back({orig_r, 0});
            while (!dfs_stack.empty()) {
                DfsStep& cur = dfs_stack.back();
                usize new_i = orig2new_row(cur.orig_i);
                const usize* children = nullptr;
                usize n_children = 0;
                if (filter(new_i)) get_children(new_i, children, n_children);
                if (!is_visited[cur.orig_i]) {
                    is_visited[cur.orig_i] = 1;
                } else {
                    cur.cur_child += 1;
                }
                while (cur.cur_child < n_children) {
                    if (!is_visited[children[cur.cur_child]]) break;
                    cur.cur_child += 1;
                }
                if (cur.cur_child < n_children) {
                    usize child = children[cur.cur_child];
                    dfs_stack.push_back({child, 0});
                } else {
                    visited.push_back(cur.orig_i);
                    dfs_stack.pop_back();
                }
            }
        }
        for (usize i : visited) {
            if (!rhs.is_nonzero[i]) {
                rhs.is_nonzero[i] = 1;
                rhs.nonzero.push_back(i);
            }
        }
    }
};

struct ScratchSpace {  // lu.rs:11-31
    ScatteredVec rhs;
    std::vector<double> dense_rhs;
    MarkNonzero mark_nonzero;
    explicit ScratchSpace(usize n = 0) : rhs(ScatteredVec::empty(n)), dense_rhs(n, 0.0) {
        mark_nonzero.is_visited.assign(n, 0);
    }
    void clear_sparse(usize size) {  // lu.rs:27-30
        rhs.clear_and_resize(size);
        mark_nonzero.clear_and_resize(size);
    }
};

// lu.rs:450-463
static inline void tri_solve_process_col(const TriangleMat& t, usize col, double* rhs) {
    double x_val = t.has_diag ? rhs[col] / t.diag[col] : rhs[col];
    rhs[col] = x_val;
    const SparseMat& m = t.nondiag;
    for (usize p = m.indptr[col]; p < m.indptr[col + 1]; ++p) rhs[m.indices[p]] -= x_val * m.data[p];
}

enum Triangle { Lower, Upper };
static void tri_solve_dense(const TriangleMat& t, Triangle tri, double* rhs) {  // lu.rs:414-429
    if (tri == Lower) {
        for (usize c = 0; c < t.cols(); ++c) tri_solve_process_col(t, c, rhs);
    } else {
        for (usize c = t.cols(); c-- > 0;) tri_solve_process_col(t, c, rhs);
    }
}

struct SolveStats {  // instrumentation only: algorithmic work of the last solve
    uint64_t tri_nnz = 0, tri_cols = 0;
};
static SolveStats g_solve_stats;

static void tri_solve_sparse(const TriangleMat& t, ScratchSpace& s) {  // lu.rs:432-448
    const SparseMat& m = t.nondiag;
    s.mark_nonzero.run(
        s.rhs,
        [&](usize col, const usize*& ch, usize& n) {
            ch = m.indices.data() + m.indptr[col];
            n = m.indptr[col + 1] - m.indptr[col];
        },
        [](usize) { return true; }, [](usize i) { return i; });
    for (usize k = s.mark_nonzero.visited.size(); k-- > 0;) {
        usize col = s.mark_nonzero.visited[k];
        tri_solve_process_col(t, col, s.rhs.values.data());
        g_solve_stats.tri_nnz += m.indptr[col + 1] - m.indptr[col];
        g_solve_stats.tri_cols += 1;
    }
}

struct LUFactors {  // lu.rs:3-9
    TriangleMat lower, upper;
    Perm row_perm, col_perm;  // always Some(..) as produced by lu_factorize
    usize nnz() const {       // lu.rs:52-54
        return lower.nondiag.nnz() + upper.nondiag.nnz() + lower.cols();
    }
    void solve_dense(double* rhs, usize n, ScratchSpace& s) const {  // lu.rs:56-77
        s.dense_rhs.resize(n, 0.0);
        for (usize i = 0; i < n; ++i) s.dense_rhs[row_perm.orig2new[i]] = rhs[i];
        tri_solve_dense(lower, Lower, s.dense_rhs.data());
        tri_solve_dense(upper, Upper, s.dense_rhs.data());
        for (usize i = 0; i < n; ++i) rhs[col_perm.new2orig[i]] = s.dense_rhs[i];
    }
    void solve(ScatteredVec& rhs, ScratchSpace& s) const {  // lu.rs:79-106
        s.rhs.clear();
        for (usize i : rhs.nonzero) {
            usize new_i = row_perm.orig2new[i];
            s.rhs.nonzero.push_back(new_i);
            s.rhs.is_nonzero[new_i] = 1;
            s.rhs.values[new_i] = rhs.values[i];
        }
        tri_solve_sparse(lower, s);
        tri_solve_sparse(upper, s);
        rhs.clear();
        for (usize i : s.rhs.nonzero) {
            usize new_i = col_perm.new2orig[i];
            rhs.nonzero.push_back(new_i);
            rhs.is_nonzero[new_i] = 1;
            rhs.values[new_i] = s.rhs.values[i];
        }
    }
    LUFactors transpose() const {  // lu.rs:108-115
        LUFactors t;
        t.lower = upper.transpose();
        t.upper = lower.transpose();
        t.row_perm = col_perm;
        t.col_perm = row_perm;
        return t;
    }
};

// lu.rs:118-304 — left-looking Gilbert–Peierls with threshold pivoting.
static LUFactors lu_factorize(usize size, const GetCol& get_col, double stability_coeff, ScratchSpace& scratch) {
    Perm col_perm = order_simple(size, get_col);  // lu.rs:140

    std::vector<usize> orig_row2elt_count(size, 0);  // lu.rs:142-147
    for (usize c = 0; c < size; ++c) {
        ColView v = get_col(c);
        for (usize p = 0; p < v.len; ++p) orig_row2elt_count[v.rows[p]] += 1;
    }

    scratch.clear_sparse(size);  // lu.rs:149

    SparseMat lower(size), upper(size);
    std::vector<double> upper_diag;
    upper_diag.reserve(size);

    std::vector<usize> new2orig_row(size), orig2new_row(size);
    for (usize i = 0; i < size; ++i) new2orig_row[i] = orig2new_row[i] = i;

    for (usize i_col = 0; i_col < size; ++i_col) {
        ColView mat_col = get_col(col_perm.new2orig[i_col]);  // lu.rs:159
        scratch.rhs.set(mat_col.rows, mat_col.vals, mat_col.len);  // lu.rs:167

        scratch.mark_nonzero.run(  // lu.rs:169-174
            scratch.rhs,
            [&](usize new_i, const usize*& ch, usize& n) {
                ch = lower.indices.data() + lower.indptr[new_i];
                n = lower.indptr[new_i + 1] - lower.indptr[new_i];
            },
            [&](usize new_i) { return new_i < i_col; },
            [&](usize orig_r) { return orig2new_row[orig_r]; });

        for (usize k = scratch.mark_nonzero.visited.size(); k-- > 0;) {  // lu.rs:179-188
            usize orig_i = scratch.mark_nonzero.visited[k];
            usize new_i = orig2new_row[orig_i];
            if (new_i < i_col) {
                double x_val = scratch.rhs.values[orig_i];
                for (usize p = lower.indptr[new_i]; p < lower.indptr[new_i + 1]; ++p)
                    scratch.rhs.values[lower.indices[p]] -= x_val * lower.data[p];
            }
        }

        // lu.rs:194-233 pivot choice
        double max_abs = 0.0;
        for (usize orig_r : scratch.rhs.nonzero) {
            if (orig2new_row[orig_r] < i_col) continue;
            double a = std::fabs(scratch.rhs.values[orig_r]);
            if (a > max_abs) max_abs = a;
        }
        if (max_abs < 1e-8) throw SingularMatrix();  // lu.rs:207-209
        if (!std::isnormal(max_abs)) throw OraclePanic("lu_factorize: max_abs not normal");  // lu.rs:211

        bool have_best = false;
        usize best_orig_r = 0, best_elt_count = 0;
        for (usize orig_r : scratch.rhs.nonzero) {
            if (orig2new_row[orig_r] < i_col) continue;
            if (std::fabs(scratch.rhs.values[orig_r]) >= stability_coeff * max_abs) {
                usize elt_count = orig_row2elt_count[orig_r];
                if (!have_best || best_elt_count > elt_count) {
                    best_orig_r = orig_r;
                    best_elt_count = elt_count;
                    have_best = true;
                }
            }
        }
        if (!have_best) throw OraclePanic("lu_factorize: no pivot");
        usize pivot_orig_r = best_orig_r;
        double pivot_val = scratch.rhs.values[pivot_orig_r];

        {  // lu.rs:237-244
            usize row = i_col;
            usize orig_row = new2orig_row[row];
            usize pivot_row = orig2new_row[pivot_orig_r];
            std::swap(new2orig_row[row], new2orig_row[pivot_row]);
            std::swap(orig2new_row[orig_row], orig2new_row[pivot_orig_r]);
        }

        for (usize orig_r : scratch.rhs.nonzero) {  // lu.rs:248-263
            double val = scratch.rhs.values[orig_r];
            if (val == 0.0) continue;
            usize new_r = orig2new_row[orig_r];
            if (new_r < i_col) {
                upper.push(new_r, val);
            } else if (new_r == i_col) {
                upper_diag.push_back(pivot_val);
            } else {
                lower.push(orig_r, val / pivot_val);
            }
        }
        upper.seal_column();
        lower.seal_column();
    }

    for (usize p = 0; p < lower.indices.size(); ++p) lower.indices[p] = orig2new_row[lower.indices[p]];  // lu.rs:270-274

    LUFactors res;
    res.lower.nondiag = std::move(lower);
    res.lower.has_diag = false;
    res.upper.nondiag = std::move(upper);
    res.upper.has_diag = true;
    res.upper.diag = std::move(upper_diag);
    res.row_perm.orig2new = std::move(orig2new_row);
    res.row_perm.new2orig = std::move(new2orig_row);
    res.col_perm = std::move(col_perm);
    return res;
}

// ---------------------------------------------------------------- solver.rs
enum ComparisonOp { OP_EQ = 0, OP_LE = 1, OP_GE = 2 };  // lib.rs:162-169

struct Constraint {
    CsVec coeffs;
    ComparisonOp op;
    double rhs;
};

struct EtaMatrices {  // solver.rs:1341-1368
    std::vector<usize> leaving_rows;
    SparseMat coeff_cols;
    usize len() const { return leaving_rows.size(); }
    void clear_and_resize(usize n) {
        leaving_rows.clear();
        coeff_cols.clear_and_resize(n);
    }
};

struct Counters {  // instrumentation added by the restatement (no counter in the reference)
    uint64_t pivots = 0, bound_flips = 0, refactors = 0;
    uint64_t primal_iters = 0, dual_iters = 0;
    uint64_t ftran = 0, btran = 0;
    uint64_t ftran_lu_nnz = 0, btran_lu_nnz = 0, eta_nnz_applied = 0, row_sweep_nnz = 0;
    double t_refactor = 0.0;
};

static double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct BasisSolver {  // solver.rs:1264-1339
    LUFactors lu_factors, lu_factors_transp;
    ScratchSpace scratch;
    EtaMatrices eta_matrices;
    ScatteredVec rhs;
    Counters* cnt = nullptr;

    void push_eta_matrix(const SparseVec& col_coeffs, usize r_leaving, double pivot_coeff) {  // solver.rs:1274-1284
        eta_matrices.leaving_rows.push_back(r_leaving);
        for (usize p = 0; p < col_coeffs.len(); ++p) {
            usize r = col_coeffs.indices[p];
            double coeff = col_coeffs.values[p];
            double val = (r == r_leaving) ? 1.0 - 1.0 / pivot_coeff : coeff / pivot_coeff;
            eta_matrices.coeff_cols.push(r, val);
        }
        eta_matrices.coeff_cols.seal_column();
    }

    void reset(const CsMat& csc, const std::vector<usize>& basic_vars) {  // solver.rs:1286-1303
        double t0 = now_s();
        scratch.clear_sparse(basic_vars.size());
        eta_matrices.clear_and_resize(basic_vars.size());
        rhs.clear_and_resize(basic_vars.size());
        GetCol get_col = [&](usize c) {
            usize v = basic_vars[c];
            usize b = csc.indptr[v], e = csc.indptr[v + 1];
            return ColView{csc.indices.data() + b, csc.data.data() + b, e - b};
        };
        try {
            lu_factors = lu_factorize(basic_vars.size(), get_col, 0.1, scratch);
        } catch (SingularMatrix&) {
            throw OraclePanic("singular basis matrix (solver.rs:1301 unwrap)");
        }
        lu_factors_transp = lu_factors.transpose();
        if (cnt) {
            cnt->refactors += 1;
            cnt->t_refactor += now_s() - t0;
        }
    }

    ScatteredVec& solve(const usize* idx, const double* val, usize k) {  // solver.rs:1305-1319
        rhs.set(idx, val, k);
        g_solve_stats = SolveStats();
        lu_factors.solve(rhs, scratch);
        for (usize e = 0; e < eta_matrices.len(); ++e) {  // Vanderbei p.139
            usize r_leaving = eta_matrices.leaving_rows[e];
            double coeff = rhs.get(r_leaving);
            const SparseMat& m = eta_matrices.coeff_cols;
            for (usize p = m.indptr[e]; p < m.indptr[e + 1]; ++p) rhs.get_mut(m.indices[p]) -= coeff * m.data[p];
        }
        if (cnt) {
            cnt->ftran += 1;
            cnt->ftran_lu_nnz += g_solve_stats.tri_nnz;
            cnt->eta_nnz_applied += eta_matrices.coeff_cols.nnz();
        }
        return rhs;
    }

    ScatteredVec& solve_transp(const usize* idx, const double* val, usize k) {  // solver.rs:1322-1338
        rhs.set(idx, val, k);
        for (usize e = eta_matrices.len(); e-- > 0;) {
            double coeff = 0.0;
            const SparseMat& m = eta_matrices.coeff_cols;
            for (usize p = m.indptr[e]; p < m.indptr[e + 1]; ++p) coeff += m.data[p] * rhs.get(m.indices[p]);
            usize r_leaving = eta_matrices.leaving_rows[e];
            rhs.get_mut(r_leaving) -= coeff;
        }
        g_solve_stats = SolveStats();
        lu_factors_transp.solve(rhs, scratch);
        if (cnt) {
            cnt->btran += 1;
            cnt->btran_lu_nnz += g_solve_stats.tri_nnz;
            cnt->eta_nnz_applied += eta_matrices.coeff_cols.nnz();
        }
        return rhs;
    }
};

struct VarState {  // solver.rs:60-64
    bool basic;
    usize idx;
};
struct NonBasicVarState {  // solver.rs:66-70
    bool at_min, at_max;
};
struct PivotElem {  // solver.rs:1256-1261
    usize row;
    double coeff, leaving_new_val;
};
struct PivotInfo {  // solver.rs:1244-1254
    usize col;
    double entering_new_val, entering_diff;
    bool has_elem;
    PivotElem elem;
};

struct PivotRecord {  // instrumentation: one per pivot when tracing is enabled
    int32_t phase;  // 0 primal, 1 dual
    int64_t col, row;  // row = -1 for a bound flip
    int64_t entering_var, leaving_var;
    double pivot_coeff, obj_after;
};

struct Solver {  // solver.rs:14-58
    usize num_vars = 0;
    std::vector<double> orig_obj_coeffs, orig_var_mins, orig_var_maxs;
    CsMat orig_constraints, orig_constraints_csc;
    std::vector<double> orig_rhs;
    bool enable_primal_steepest_edge = false, enable_dual_steepest_edge = false;
    bool is_primal_feasible = false, is_dual_feasible = false;
    std::vector<VarState> var_states;
    BasisSolver basis_solver;
    std::vector<usize> basic_vars;
    std::vector<double> basic_var_vals, basic_var_mins, basic_var_maxs, dual_edge_sq_norms;
    std::vector<usize> nb_vars;
    std::vector<double> nb_var_obj_coeffs, nb_var_vals;
    std::vector<NonBasicVarState> nb_var_states;
    std::vector<uint8_t> nb_var_is_fixed;
    std::vector<double> primal_edge_sq_norms;
    double cur_obj_val = 0.0;
    SparseVec col_coeffs;
    std::vector<double> sq_norms_update_helper;
    SparseVec inv_basis_row_coeffs;
    ScatteredVec row_coeffs;

    // instrumentation (not in the reference)
    Counters cnt;
    int64_t pivot_budget = -1;  // <0: unlimited; else stop loops after this many more pivots
    bool budget_exhausted = false;
    // test instrumentation (not in the reference): with capture on, the two extra solves of a pivot are kept so that
    // the per-stage differential tests can compare them (v = B^-T alpha_q, solver.rs:1114; tau = B^-1 rho, solver.rs:1157)
    bool capture = false;
    std::vector<double> dbg_v, dbg_tau, dbg_rho;
    bool resume_in_optimize = false;
    bool trace = false;
    std::vector<PivotRecord> trace_log;

    Solver() {}
    Solver(const Solver& o) { *this = o; }
    Solver& operator=(const Solver& o) = default;
    void fix_ptrs() { basis_solver.cnt = &cnt; }

    usize num_constraints() const { return orig_constraints.outer(); }          // solver.rs:462-464
    usize num_total_vars() const { return num_vars + num_constraints(); }       // solver.rs:466-468

    // solver.rs:108-369
    void try_new(const std::vector<double>& obj_coeffs, const std::vector<double>& var_mins,
                 const std::vector<double>& var_maxs, const std::vector<Constraint>& constraints) {
        fix_ptrs();
        const bool enable_steepest_edge = true;  // solver.rs:114
        num_vars = obj_coeffs.size();
        orig_var_mins = var_mins;
        orig_var_maxs = var_maxs;
        double obj_val = 0.0;
        is_dual_feasible = true;

        for (usize v = 0; v < num_vars; ++v) {  // solver.rs:133-187
            double min = orig_var_mins[v], max = orig_var_maxs[v];
            if (min > max) throw LpFail{LP_INFEASIBLE};
            var_states.push_back({false, nb_vars.size()});
            nb_vars.push_back(v);
            double init_val;
            if (min == max) {
                init_val = min;
            } else if (std::isinf(min) && std::isinf(max)) {
                if (obj_coeffs[v] != 0.0) is_dual_feasible = false;
                init_val = 0.0;
            } else if (obj_coeffs[v] > 0.0) {
                if (std::isfinite(min)) {
                    init_val = min;
                } else {
                    is_dual_feasible = false;
                    init_val = max;
                }
            } else if (obj_coeffs[v] < 0.0) {
                if (std::isfinite(max)) {
                    init_val = max;
                } else {
                    is_dual_feasible = false;
                    init_val = min;
                }
            } else if (std::isfinite(min)) {
                init_val = min;
            } else {
                init_val = max;
            }
            nb_var_vals.push_back(init_val);
            obj_val += init_val * obj_coeffs[v];
            nb_var_states.push_back({init_val == min, init_val == max});
        }

        std::vector<const CsVec*> constraint_coeffs;
        for (const Constraint& c : constraints) {  // solver.rs:198-239
            double rhs = c.rhs;
            if (c.coeffs.indices.empty()) {
                bool taut = (c.op == OP_EQ) ? (0.0 == rhs) : (c.op == OP_LE) ? (0.0 <= rhs) : (0.0 >= rhs);
                if (taut) continue;
                throw LpFail{LP_INFEASIBLE};
            }
            constraint_coeffs.push_back(&c.coeffs);
            orig_rhs.push_back(rhs);
            double smin, smax;
            if (c.op == OP_LE) { smin = 0.0; smax = INF; }
            else if (c.op == OP_GE) { smin = -INF; smax = 0.0; }
            else { smin = 0.0; smax = 0.0; }
            orig_var_mins.push_back(smin);
            orig_var_maxs.push_back(smax);
            basic_var_mins.push_back(smin);
            basic_var_maxs.push_back(smax);
            usize cur_slack_var = var_states.size();
            var_states.push_back({true, basic_vars.size()});
            basic_vars.push_back(cur_slack_var);
            double lhs_val = 0.0;
            for (usize p = 0; p < c.coeffs.indices.size(); ++p) lhs_val += c.coeffs.data[p] * nb_var_vals[c.coeffs.indices[p]];
            basic_var_vals.push_back(rhs - lhs_val);
        }

        usize n_constr = constraint_coeffs.size();
        usize n_total = num_vars + n_constr;
        orig_obj_coeffs = obj_coeffs;
        orig_obj_coeffs.resize(n_total, 0.0);

        orig_constraints = CsMat();  // solver.rs:247-253
        orig_constraints.inner_dim = n_total;
        for (usize s = 0; s < n_constr; ++s) {
            CsVec coeffs = into_resized(*constraint_coeffs[s], n_total);
            coeffs.append(num_vars + s, 1.0);
            orig_constraints.append_outer(coeffs.indices.data(), coeffs.data.data(), coeffs.indices.size());
        }
        orig_constraints_csc = orig_constraints.to_other_storage();

        is_primal_feasible = true;  // solver.rs:255-259
        for (usize r = 0; r < basic_var_vals.size(); ++r)
            if (!(basic_var_vals[r] >= basic_var_mins[r] && basic_var_vals[r] <= basic_var_maxs[r])) {
                is_primal_feasible = false;
                break;
            }
        bool need_artificial_obj = !is_primal_feasible && !is_dual_feasible;  // solver.rs:261

        enable_dual_steepest_edge = enable_steepest_edge;  // solver.rs:263-268
        if (enable_dual_steepest_edge) dual_edge_sq_norms.assign(basic_vars.size(), 1.0);
        enable_primal_steepest_edge = enable_steepest_edge && !is_dual_feasible;  // solver.rs:272
        if (enable_primal_steepest_edge) sq_norms_update_helper.assign(n_total - n_constr, 0.0);

        for (usize i = 0; i < nb_vars.size(); ++i) {  // solver.rs:281-300
            usize var = nb_vars[i];
            const NonBasicVarState& st = nb_var_states[i];
            if (need_artificial_obj) {
                double coeff = (st.at_min && !st.at_max) ? 1.0 : (st.at_max && !st.at_min) ? -1.0 : 0.0;
                nb_var_obj_coeffs.push_back(coeff);
            } else {
                nb_var_obj_coeffs.push_back(orig_obj_coeffs[var]);
            }
            if (enable_primal_steepest_edge) {
                double s = 0.0;  // sprs squared_l2_norm
                for (usize p = orig_constraints_csc.indptr[var]; p < orig_constraints_csc.indptr[var + 1]; ++p)
                    s += orig_constraints_csc.data[p] * orig_constraints_csc.data[p];
                primal_edge_sq_norms.push_back(s + 1.0);
            }
        }
        cur_obj_val = need_artificial_obj ? 0.0 : obj_val;  // solver.rs:302

        basis_solver.scratch = ScratchSpace(n_constr);  // solver.rs:304-317
        basis_solver.eta_matrices.coeff_cols = SparseMat(n_constr);
        basis_solver.rhs = ScatteredVec::empty(n_constr);
        GetCol get_col = [&](usize c) {
            usize v = basic_vars[c];
            usize b = orig_constraints_csc.indptr[v], e = orig_constraints_csc.indptr[v + 1];
            return ColView{orig_constraints_csc.indices.data() + b, orig_constraints_csc.data.data() + b, e - b};
        };
        try {
            basis_solver.lu_factors = lu_factorize(basic_vars.size(), get_col, 0.1, basis_solver.scratch);
        } catch (SingularMatrix&) {
            throw OraclePanic("singular initial basis (solver.rs:316 unwrap)");
        }
        basis_solver.lu_factors_transp = basis_solver.lu_factors.transpose();
        nb_var_is_fixed.assign(nb_vars.size(), 0);
        row_coeffs = ScatteredVec::empty(n_total - n_constr);
    }

    double get_value(usize var) const {  // solver.rs:371-376
        const VarState& s = var_states[var];
        return s.basic ? basic_var_vals[s.idx] : nb_var_vals[s.idx];
    }

    void fix_var(usize var, double val) {  // solver.rs:378-415
        if (val < orig_var_mins[var] || val > orig_var_maxs[var]) throw LpFail{LP_INFEASIBLE};
        usize col;
        if (var_states[var].basic) {
            usize row = var_states[var].idx;
            calc_row_coeffs(row);
            PivotInfo pi = choose_entering_col_dual(row, val);
            calc_col_coeffs(pi.col);
            pivot(pi, 1);
            col = pi.col;
        } else {
            col = var_states[var].idx;
            calc_col_coeffs(col);
            double diff = val - nb_var_vals[col];
            for (usize p = 0; p < col_coeffs.len(); ++p) basic_var_vals[col_coeffs.indices[p]] -= diff * col_coeffs.values[p];
            cur_obj_val += diff * nb_var_obj_coeffs[col];
            nb_var_vals[col] = val;
        }
        nb_var_states[col] = {true, true};
        nb_var_is_fixed[col] = 1;
        is_primal_feasible = false;
        restore_feasibility();
    }

    bool unfix_var(usize var) {  // solver.rs:418-438
        if (!var_states[var].basic) {
            usize col = var_states[var].idx;
            bool was = nb_var_is_fixed[col];
            nb_var_is_fixed[col] = 0;
            if (!was) return false;
            double cur_val = nb_var_vals[col];
            nb_var_states[col] = {cur_val == orig_var_mins[var], cur_val == orig_var_maxs[var]};
            is_dual_feasible = false;
            try {
                optimize();
            } catch (LpFail&) {
                throw OraclePanic("unfix_var: optimize failed (solver.rs:433 unwrap)");
            }
            return true;
        }
        return false;
    }

    void add_gomory_cut(usize var) {  // solver.rs:440-460
        if (!var_states[var].basic) throw OraclePanic("add_gomory_cut: var is not basic (solver.rs:458)");
        usize row = var_states[var].idx;
        calc_row_coeffs(row);
        std::vector<usize> idx;
        std::vector<double> val;
        for (usize col : row_coeffs.nonzero) {
            double coeff = row_coeffs.values[col];
            idx.push_back(nb_vars[col]);
            val.push_back(std::floor(coeff) - coeff);
        }
        double cut_bound = std::floor(basic_var_vals[row]) - basic_var_vals[row];
        usize n_total = num_total_vars();
        add_constraint(CsVec::make(n_total, idx, val), OP_LE, cut_bound);
    }

    void initial_solve() {  // solver.rs:470-485
        if (!is_primal_feasible) restore_feasibility();
        if (budget_exhausted) return;
        if (!is_dual_feasible) {
            if (!resume_in_optimize) recalc_obj_coeffs();  // a budget resume must not recompute d
            resume_in_optimize = true;
            optimize();
        }
        if (budget_exhausted) return;
        resume_in_optimize = false;
        enable_primal_steepest_edge = false;
    }

    bool take_budget() {  // instrumentation: fixed-pivot-budget protocol
        if (pivot_budget < 0) return true;
        if (pivot_budget == 0) {
            budget_exhausted = true;
            return false;
        }
        pivot_budget -= 1;
        return true;
    }

    void optimize() {  // solver.rs:487-511
        for (;;) {
            if (!take_budget()) return;
            PivotInfo pi;
            if (choose_pivot(pi)) {
                cnt.primal_iters += 1;
                pivot(pi, 0);
            } else {
                break;
            }
        }
        is_dual_feasible = true;
    }

    void restore_feasibility() {  // solver.rs:513-547
        for (;;) {
            if (!take_budget()) return;
            usize row;
            double leaving_new_val;
            if (choose_pivot_row_dual(row, leaving_new_val)) {
                calc_row_coeffs(row);
                PivotInfo pi = choose_entering_col_dual(row, leaving_new_val);
                calc_col_coeffs(pi.col);
                cnt.dual_iters += 1;
                pivot(pi, 1);
            } else {
                break;
            }
        }
        is_primal_feasible = true;
    }

    void add_constraint(CsVec coeffs, ComparisonOp op, double rhs) {  // solver.rs:549-634
        if (!is_primal_feasible || !is_dual_feasible) throw OraclePanic("add_constraint: model not solved (solver.rs:555-556)");
        if (coeffs.indices.empty()) {
            bool taut = (op == OP_EQ) ? (0.0 == rhs) : (op == OP_LE) ? (0.0 <= rhs) : (0.0 >= rhs);
            if (taut) return;
            throw LpFail{LP_INFEASIBLE};
        }
        usize slack_var = num_total_vars();
        double smin, smax;
        if (op == OP_LE) { smin = 0.0; smax = INF; }
        else if (op == OP_GE) { smin = -INF; smax = 0.0; }
        else { smin = 0.0; smax = 0.0; }
        orig_obj_coeffs.push_back(0.0);
        orig_var_mins.push_back(smin);
        orig_var_maxs.push_back(smax);
        var_states.push_back({true, basic_vars.size()});
        basic_vars.push_back(slack_var);
        basic_var_mins.push_back(smin);
        basic_var_maxs.push_back(smax);

        double lhs_val = 0.0;  // solver.rs:587-595
        for (usize p = 0; p < coeffs.indices.size(); ++p) {
            const VarState& s = var_states[coeffs.indices[p]];
            double val = s.basic ? basic_var_vals[s.idx] : nb_var_vals[s.idx];
            lhs_val += val * coeffs.data[p];
        }
        basic_var_vals.push_back(rhs - lhs_val);

        usize new_n_total = num_total_vars() + 1;  // solver.rs:597-610: full CSR rebuild + to_csc
        CsMat new_csr;
        new_csr.inner_dim = new_n_total;
        for (usize r = 0; r < orig_constraints.outer(); ++r) {
            usize b = orig_constraints.indptr[r], e = orig_constraints.indptr[r + 1];
            new_csr.append_outer(orig_constraints.indices.data() + b, orig_constraints.data.data() + b, e - b);
        }
        coeffs = into_resized(coeffs, new_n_total);
        coeffs.append(slack_var, 1.0);
        new_csr.append_outer(coeffs.indices.data(), coeffs.data.data(), coeffs.indices.size());
        orig_rhs.push_back(rhs);
        orig_constraints = std::move(new_csr);
        orig_constraints_csc = orig_constraints.to_other_storage();

        basis_solver.reset(orig_constraints_csc, basic_vars);  // solver.rs:612-613

        if (enable_primal_steepest_edge || enable_dual_steepest_edge) {  // solver.rs:615-630
            calc_row_coeffs(num_constraints() - 1);
            if (enable_primal_steepest_edge)
                for (usize c : row_coeffs.nonzero) primal_edge_sq_norms[c] += row_coeffs.values[c] * row_coeffs.values[c];
            if (enable_dual_steepest_edge) dual_edge_sq_norms.push_back(inv_basis_row_coeffs.sq_norm());
        }
        is_primal_feasible = false;
        restore_feasibility();
    }

    void calc_primal_infeasibility(usize& n, double& inf) const {  // solver.rs:637-655
        n = 0;
        inf = 0.0;
        for (usize r = 0; r < basic_var_vals.size(); ++r) {
            double val = basic_var_vals[r], mn = basic_var_mins[r], mx = basic_var_maxs[r];
            if (val < mn - EPS) { n += 1; inf += mn - val; }
            else if (val > mx + EPS) { n += 1; inf += val - mx; }
        }
    }
    void calc_dual_infeasibility(usize& n, double& inf) const {  // solver.rs:658-668
        n = 0;
        inf = 0.0;
        for (usize c = 0; c < nb_var_obj_coeffs.size(); ++c) {
            double d = nb_var_obj_coeffs[c];
            const NonBasicVarState& s = nb_var_states[c];
            if (!(s.at_min && d > -EPS) && !(s.at_max && d < EPS)) { n += 1; inf += std::fabs(d); }
        }
    }

    void calc_col_coeffs(usize c_var) {  // solver.rs:671-677
        usize var = nb_vars[c_var];
        usize b = orig_constraints_csc.indptr[var], e = orig_constraints_csc.indptr[var + 1];
        basis_solver.solve(orig_constraints_csc.indices.data() + b, orig_constraints_csc.data.data() + b, e - b)
            .to_sparse_vec(col_coeffs);
    }

    void calc_row_coeffs(usize r_constr) {  // solver.rs:680-693
        double one = 1.0;
        basis_solver.solve_transp(&r_constr, &one, 1).to_sparse_vec(inv_basis_row_coeffs);
        row_coeffs.clear_and_resize(nb_vars.size());
        for (usize p = 0; p < inv_basis_row_coeffs.len(); ++p) {
            usize r = inv_basis_row_coeffs.indices[p];
            double coeff = inv_basis_row_coeffs.values[p];
            usize b = orig_constraints.indptr[r], e = orig_constraints.indptr[r + 1];
            cnt.row_sweep_nnz += e - b;
            for (usize q = b; q < e; ++q) {
                const VarState& s = var_states[orig_constraints.indices[q]];
                if (!s.basic) row_coeffs.get_mut(s.idx) += orig_constraints.data[q] * coeff;
            }
        }
    }

    // solver.rs:752-771 get_leaving_var_step
    double leaving_step(usize r, double coeff, bool entering_diff_sign) const {
        double val = basic_var_vals[r];
        if ((entering_diff_sign && coeff < 0.0) || (!entering_diff_sign && coeff > 0.0)) {
            double mx = basic_var_maxs[r];
            return val < mx ? mx - val : 0.0;
        } else {
            double mn = basic_var_mins[r];
            return val > mn ? val - mn : 0.0;
        }
    }

    bool choose_pivot(PivotInfo& out) {  // solver.rs:695-853; false = optimal
        bool have = false;
        usize entering_c = 0;
        double best_score = -INF;
        for (usize col = 0; col < nb_var_obj_coeffs.size(); ++col) {  // solver.rs:696-732
            double d = nb_var_obj_coeffs[col];
            const NonBasicVarState& s = nb_var_states[col];
            if ((s.at_min && d > -EPS) || (s.at_max && d < EPS)) continue;
            double score = enable_primal_steepest_edge ? d * d / primal_edge_sq_norms[col] : std::fabs(d);
            if (score > best_score) {
                have = true;
                entering_c = col;
                best_score = score;
            }
        }
        if (!have) return false;

        double entering_cur_val = nb_var_vals[entering_c];  // solver.rs:741-748
        bool entering_diff_sign = nb_var_obj_coeffs[entering_c] < 0.0;
        double entering_other_val = entering_diff_sign ? orig_var_maxs[nb_vars[entering_c]] : orig_var_mins[nb_vars[entering_c]];

        calc_col_coeffs(entering_c);  // solver.rs:750

        double max_step = std::fabs(entering_other_val - entering_cur_val);  // solver.rs:782-795
        for (usize p = 0; p < col_coeffs.len(); ++p) {
            double coeff = col_coeffs.values[p];
            double coeff_abs = std::fabs(coeff);
            if (coeff_abs < EPS) continue;
            double cur_step = (leaving_step(col_coeffs.indices[p], coeff, entering_diff_sign) + EPS) / coeff_abs;
            if (cur_step < max_step) max_step = cur_step;
        }

        bool have_r = false;  // solver.rs:800-823
        usize leaving_r = 0;
        double leaving_new_val = 0.0, pivot_coeff_abs = -INF, pivot_coeff = 0.0;
        for (usize p = 0; p < col_coeffs.len(); ++p) {
            usize r = col_coeffs.indices[p];
            double coeff = col_coeffs.values[p];
            double coeff_abs = std::fabs(coeff);
            if (coeff_abs < EPS) continue;
            double cur_step = leaving_step(r, coeff, entering_diff_sign) / coeff_abs;
            if (cur_step <= max_step && coeff_abs > pivot_coeff_abs) {
                have_r = true;
                leaving_r = r;
                leaving_new_val = ((entering_diff_sign && coeff < 0.0) || (!entering_diff_sign && coeff > 0.0))
                                      ? basic_var_maxs[r] : basic_var_mins[r];
                pivot_coeff = coeff;
                pivot_coeff_abs = coeff_abs;
            }
        }

        if (have_r) {  // solver.rs:825-840
            calc_row_coeffs(leaving_r);
            double entering_diff = (basic_var_vals[leaving_r] - leaving_new_val) / pivot_coeff;
            out.col = entering_c;
            out.entering_new_val = entering_cur_val + entering_diff;
            out.entering_diff = entering_diff;
            out.has_elem = true;
            out.elem = {leaving_r, pivot_coeff, leaving_new_val};
            return true;
        }
        if (std::isinf(entering_other_val)) throw LpFail{LP_UNBOUNDED};  // solver.rs:842-844
        out.col = entering_c;  // solver.rs:846-851
        out.entering_new_val = entering_other_val;
        out.entering_diff = entering_other_val - entering_cur_val;
        out.has_elem = false;
        return true;
    }

    bool choose_pivot_row_dual(usize& row, double& new_val) const {  // solver.rs:855-917
        bool have = false;
        usize leaving_r = 0;
        double max_score = -INF;
        for (usize r = 0; r < basic_var_vals.size(); ++r) {
            double val = basic_var_vals[r], mn = basic_var_mins[r], mx = basic_var_maxs[r];
            double infeas;
            if (val < mn - EPS) infeas = mn - val;
            else if (val > mx + EPS) infeas = val - mx;
            else continue;
            double score = enable_dual_steepest_edge ? infeas * infeas / dual_edge_sq_norms[r] : infeas;
            if (score > max_score) {
                have = true;
                leaving_r = r;
                max_score = score;
            }
        }
        if (!have) return false;
        double val = basic_var_vals[leaving_r];
        if (val < basic_var_mins[leaving_r]) new_val = basic_var_mins[leaving_r];
        else if (val > basic_var_maxs[leaving_r]) new_val = basic_var_maxs[leaving_r];
        else throw OraclePanic("choose_pivot_row_dual: unreachable (solver.rs:913)");
        row = leaving_r;
        return true;
    }

    static double clamp_obj_coeff(double d, const NonBasicVarState& s) {  // solver.rs:927-935
        if (s.at_min && d < 0.0) d = 0.0;
        if (s.at_max && d > 0.0) d = 0.0;
        return d;
    }
    static bool is_eligible_var(double coeff, const NonBasicVarState& s, bool leaving_diff_sign) {  // solver.rs:937-951
        bool entering_diff_sign;
        if (coeff >= EPS) entering_diff_sign = !leaving_diff_sign;
        else if (coeff <= -EPS) entering_diff_sign = leaving_diff_sign;
        else return false;
        return entering_diff_sign ? !s.at_max : !s.at_min;
    }

    PivotInfo choose_entering_col_dual(usize row, double leaving_new_val) const {  // solver.rs:919-1021
        bool leaving_diff_sign = leaving_new_val > basic_var_vals[row];
        double max_step = INF;  // solver.rs:962-974
        for (usize c : row_coeffs.nonzero) {
            double coeff = row_coeffs.values[c];
            const NonBasicVarState& s = nb_var_states[c];
            if (!is_eligible_var(coeff, s, leaving_diff_sign)) continue;
            double d = clamp_obj_coeff(nb_var_obj_coeffs[c], s);
            double cur_step = (std::fabs(d) + EPS) / std::fabs(coeff);
            if (cur_step < max_step) max_step = cur_step;
        }
        bool have = false;  // solver.rs:979-1002
        usize entering_c = 0;
        double pivot_coeff_abs = -INF, pivot_coeff = 0.0;
        for (usize c : row_coeffs.nonzero) {
            double coeff = row_coeffs.values[c];
            const NonBasicVarState& s = nb_var_states[c];
            if (!is_eligible_var(coeff, s, leaving_diff_sign)) continue;
            double d = clamp_obj_coeff(nb_var_obj_coeffs[c], s);
            double cur_step = std::fabs(d) / std::fabs(coeff);
            if (cur_step <= max_step) {
                double coeff_abs = std::fabs(coeff);
                if (coeff_abs > pivot_coeff_abs) {
                    have = true;
                    entering_c = c;
                    pivot_coeff_abs = coeff_abs;
                    pivot_coeff = coeff;
                }
            }
        }
        if (!have) throw LpFail{LP_INFEASIBLE};  // solver.rs:1019
        PivotInfo pi;
        pi.entering_diff = (basic_var_vals[row] - leaving_new_val) / pivot_coeff;
        pi.entering_new_val = nb_var_vals[entering_c] + pi.entering_diff;
        pi.col = entering_c;
        pi.has_elem = true;
        pi.elem = {row, pivot_coeff, leaving_new_val};
        return pi;
    }

    void pivot(const PivotInfo& pi, int phase) {  // solver.rs:1023-1104
        cur_obj_val += nb_var_obj_coeffs[pi.col] * pi.entering_diff;  // solver.rs:1027
        usize entering_var = nb_vars[pi.col];

        if (!pi.has_elem) {  // solver.rs:1031-1042 bound flip
            nb_var_vals[pi.col] = pi.entering_new_val;
            for (usize p = 0; p < col_coeffs.len(); ++p) basic_var_vals[col_coeffs.indices[p]] -= pi.entering_diff * col_coeffs.values[p];
            nb_var_states[pi.col].at_min = pi.entering_new_val == orig_var_mins[entering_var];
            nb_var_states[pi.col].at_max = pi.entering_new_val == orig_var_maxs[entering_var];
            cnt.bound_flips += 1;
            if (trace) trace_log.push_back({phase, (int64_t)pi.col, -1, (int64_t)entering_var, -1, 0.0, cur_obj_val});
            return;
        }
        const PivotElem& pe = pi.elem;
        double pivot_coeff = pe.coeff;

        for (usize p = 0; p < col_coeffs.len(); ++p) {  // solver.rs:1049-1055
            usize r = col_coeffs.indices[p];
            if (r == pe.row) basic_var_vals[r] = pi.entering_new_val;
            else basic_var_vals[r] -= pi.entering_diff * col_coeffs.values[p];
        }
        basic_var_mins[pe.row] = orig_var_mins[entering_var];  // solver.rs:1057-1058
        basic_var_maxs[pe.row] = orig_var_maxs[entering_var];

        if (enable_dual_steepest_edge) update_dual_sq_norms(pe.row, pivot_coeff);  // solver.rs:1060-1062

        usize leaving_var = basic_vars[pe.row];  // solver.rs:1066-1071
        nb_var_vals[pi.col] = pe.leaving_new_val;
        nb_var_states[pi.col].at_min = pe.leaving_new_val == orig_var_mins[leaving_var];
        nb_var_states[pi.col].at_max = pe.leaving_new_val == orig_var_maxs[leaving_var];

        double pivot_obj = nb_var_obj_coeffs[pi.col] / pivot_coeff;  // solver.rs:1073-1080
        for (usize c : row_coeffs.nonzero) {
            if (c == pi.col) nb_var_obj_coeffs[c] = -pivot_obj;
            else nb_var_obj_coeffs[c] -= pivot_obj * row_coeffs.values[c];
        }

        if (enable_primal_steepest_edge) update_primal_sq_norms(pi.col, pivot_coeff);  // solver.rs:1082-1084

        basic_vars[pe.row] = entering_var;  // solver.rs:1088-1091
        var_states[entering_var] = {true, pe.row};
        nb_vars[pi.col] = leaving_var;
        var_states[leaving_var] = {false, pi.col};

        cnt.pivots += 1;
        if (trace) trace_log.push_back({phase, (int64_t)pi.col, (int64_t)pe.row, (int64_t)entering_var, (int64_t)leaving_var, pivot_coeff, cur_obj_val});

        usize eta_nnz = basis_solver.eta_matrices.coeff_cols.nnz();  // solver.rs:1096-1103
        if (eta_nnz < basis_solver.lu_factors.nnz()) basis_solver.push_eta_matrix(col_coeffs, pe.row, pivot_coeff);
        else basis_solver.reset(orig_constraints_csc, basic_vars);
    }

    void update_primal_sq_norms(usize entering_col, double pivot_coeff) {  // solver.rs:1106-1151 (Forrest–Goldfarb)
        ScatteredVec& tmp = basis_solver.solve_transp(col_coeffs.indices.data(), col_coeffs.values.data(), col_coeffs.len());
        if (capture) {
            dbg_v.assign(num_constraints(), 0.0);
            for (usize r : tmp.nonzero) dbg_v[r] = tmp.values[r];
        }
        for (usize r : tmp.nonzero) {  // solver.rs:1117-1123
            for (usize q = orig_constraints.indptr[r]; q < orig_constraints.indptr[r + 1]; ++q) {
                const VarState& s = var_states[orig_constraints.indices[q]];
                if (!s.basic) sq_norms_update_helper[s.idx] = 0.0;
            }
        }
        for (usize r : tmp.nonzero) {  // solver.rs:1126-1132
            double coeff = tmp.values[r];
            usize b = orig_constraints.indptr[r], e = orig_constraints.indptr[r + 1];
            cnt.row_sweep_nnz += 2 * (e - b);
            for (usize q = b; q < e; ++q) {
                const VarState& s = var_states[orig_constraints.indices[q]];
                if (!s.basic) sq_norms_update_helper[s.idx] += orig_constraints.data[q] * coeff;
            }
        }
        double pivot_sq_norm = col_coeffs.sq_norm() + 1.0;  // solver.rs:1136
        double pivot_coeff_sq = pivot_coeff * pivot_coeff;
        for (usize c : row_coeffs.nonzero) {  // solver.rs:1140-1150
            double r_coeff = row_coeffs.values[c];
            if (c == entering_col) {
                primal_edge_sq_norms[c] = pivot_sq_norm / pivot_coeff_sq;
            } else {
                primal_edge_sq_norms[c] += -2.0 * r_coeff * sq_norms_update_helper[c] / pivot_coeff
                                           + pivot_sq_norm * r_coeff * r_coeff / pivot_coeff_sq;
            }
            if (!std::isfinite(primal_edge_sq_norms[c])) throw OraclePanic("primal sq norm not finite (solver.rs:1149)");
        }
    }

    void update_dual_sq_norms(usize leaving_row, double pivot_coeff) {  // solver.rs:1153-1174
        ScatteredVec& tau = basis_solver.solve(inv_basis_row_coeffs.indices.data(), inv_basis_row_coeffs.values.data(),
                                               inv_basis_row_coeffs.len());
        if (capture) {
            dbg_tau.assign(num_constraints(), 0.0);
            for (usize r : tau.nonzero) dbg_tau[r] = tau.values[r];
            dbg_rho.assign(num_constraints(), 0.0);
            for (usize p = 0; p < inv_basis_row_coeffs.len(); ++p) dbg_rho[inv_basis_row_coeffs.indices[p]] = inv_basis_row_coeffs.values[p];
        }
        double pivot_sq_norm = inv_basis_row_coeffs.sq_norm();
        double pivot_coeff_sq = pivot_coeff * pivot_coeff;
        for (usize p = 0; p < col_coeffs.len(); ++p) {
            usize r = col_coeffs.indices[p];
            double col_coeff = col_coeffs.values[p];
            if (r == leaving_row) {
                dual_edge_sq_norms[r] = pivot_sq_norm / pivot_coeff_sq;
            } else {
                dual_edge_sq_norms[r] += -2.0 * col_coeff * tau.get(r) / pivot_coeff
                                         + pivot_sq_norm * col_coeff * col_coeff / pivot_coeff_sq;
            }
            if (!std::isfinite(dual_edge_sq_norms[r])) throw OraclePanic("dual sq norm not finite (solver.rs:1172)");
        }
    }

    void recalc_obj_coeffs() {  // solver.rs:1199-1231
        if (basis_solver.eta_matrices.len() > 0) basis_solver.reset(orig_constraints_csc, basic_vars);
        std::vector<double> multipliers(num_constraints(), 0.0);
        for (usize c = 0; c < basic_vars.size(); ++c) multipliers[c] = orig_obj_coeffs[basic_vars[c]];
        basis_solver.lu_factors_transp.solve_dense(multipliers.data(), multipliers.size(), basis_solver.scratch);
        nb_var_obj_coeffs.clear();
        for (usize var : nb_vars) {
            double dot = 0.0;
            for (usize p = orig_constraints_csc.indptr[var]; p < orig_constraints_csc.indptr[var + 1]; ++p)
                dot += orig_constraints_csc.data[p] * multipliers[orig_constraints_csc.indices[p]];
            nb_var_obj_coeffs.push_back(orig_obj_coeffs[var] - dot);
        }
        cur_obj_val = 0.0;
        for (usize r = 0; r < basic_vars.size(); ++r) cur_obj_val += orig_obj_coeffs[basic_vars[r]] * basic_var_vals[r];
        for (usize c = 0; c < nb_vars.size(); ++c) cur_obj_val += orig_obj_coeffs[nb_vars[c]] * nb_var_vals[c];
    }
};

// ------------------------------------------------------------------- lib.rs
struct Problem {  // lib.rs:193-305
    int direction = 0;  // 0 Minimize, 1 Maximize
    std::vector<double> obj_coeffs, var_mins, var_maxs;
    std::vector<Constraint> constraints;
    usize add_var(double obj, double mn, double mx) {  // lib.rs:233-243
        usize v = obj_coeffs.size();
        obj_coeffs.push_back(direction == 1 ? -obj : obj);
        var_mins.push_back(mn);
        var_maxs.push_back(mx);
        return v;
    }
    void add_constraint(std::vector<usize> idx, std::vector<double> val, ComparisonOp op, double rhs) {  // lib.rs:276-283
        constraints.push_back({CsVec::make(obj_coeffs.size(), idx, val), op, rhs});
    }
};

struct Solution {  // lib.rs:313-424
    int direction = 0;
    usize num_vars = 0;
    Solver solver;
    Solution() {}
    Solution(const Solution& o) : direction(o.direction), num_vars(o.num_vars), solver(o.solver) { solver.fix_ptrs(); }
    double objective() const { return direction == 1 ? -solver.cur_obj_val : solver.cur_obj_val; }  // lib.rs:334-339
};

// ------------------------------------------------------------------- mps.rs
struct MpsError : std::runtime_error {
    explicit MpsError(const std::string& s) : std::runtime_error(s) {}
};

struct MpsFile {  // mps.rs:7-16
    std::string problem_name;
    std::vector<std::string> var_names;  // index = Variable
    std::map<std::string, usize> variables;
    Problem problem;
};

static MpsFile mps_parse(const std::string& text, int direction) {  // mps.rs:39-328
    std::vector<std::string> raw;
    {
        usize b = 0;
        while (b < text.size()) {
            usize e = text.find('\n', b);
            if (e == std::string::npos) e = text.size();
            raw.push_back(text.substr(b, e - b));
            b = e + 1;
        }
    }
    usize pos = 0, idx = 0;
    std::string cur;
    auto err = [&](const std::string& m) { return MpsError("line " + std::to_string(idx) + ": " + m); };
    auto to_next = [&]() {  // mps.rs:338-356
        for (;;) {
            idx += 1;
            if (pos >= raw.size()) { cur.clear(); return; }
            cur = raw[pos++];
            if (!cur.empty() && cur[0] == '*') continue;
            usize len = cur.size();
            while (len > 0 && isspace((unsigned char)cur[len - 1])) len--;
            if (len != 0) { cur.resize(len); return; }
        }
    };
    auto tokens_of = [&](const std::string& line) {
        std::vector<std::string> t;
        std::istringstream ss(line);
        std::string w;
        while (ss >> w) t.push_back(w);
        return t;
    };
    auto parse_f64 = [&](const std::string& s) {  // mps.rs:389-400
        char* end = nullptr;
        double v = strtod(s.c_str(), &end);
        if (end == s.c_str() || *end != 0) throw MpsError("line " + std::to_string(idx) + ": couldn't parse float from string: `" + s + "`");
        return v;
    };
    auto need = [&](const std::vector<std::string>& t, usize i) -> const std::string& {
        if (i >= t.size()) throw MpsError("line " + std::to_string(idx) + ": unexpected end of line");
        return t[i];
    };
    auto kv_pairs = [&](const std::vector<std::string>& t, usize from) {  // mps.rs:402-433
        std::vector<std::pair<std::string, double>> kv;
        kv.push_back({need(t, from), 0.0});
        kv[0].second = parse_f64(need(t, from + 1));
        if (from + 2 < t.size()) {
            kv.push_back({t[from + 2], 0.0});
            kv[1].second = parse_f64(need(t, from + 3));
        }
        return kv;
    };
    auto starts_space = [&]() { return !cur.empty() && cur[0] == ' '; };

    MpsFile out;
    to_next();  // mps.rs:50-57
    {
        auto t = tokens_of(cur);
        if (t.empty()) throw MpsError("line " + std::to_string(idx) + ": unexpected end of line");
        if (t[0] != "NAME") throw err("expected NAME section");
        out.problem_name = t.size() > 1 ? t[1] : "";
    }
    struct ConstraintDef {
        std::vector<usize> vars;
        std::vector<double> coeffs;
        ComparisonOp op;
        double rhs, range;
    };
    bool have_obj = false;
    std::string obj_name;
    std::set<std::string> free_rows;
    std::vector<ConstraintDef> constraints;
    std::map<std::string, usize> constr_name2idx;
    to_next();  // mps.rs:71-114
    if (cur != "ROWS") throw err("expected ROWS section");
    for (;;) {
        to_next();
        if (!starts_space()) break;
        auto t = tokens_of(cur);
        const std::string& row_type = need(t, 0);
        const std::string& name = need(t, 1);
        ComparisonOp op;
        if (row_type == "N") {
            if (!have_obj) { have_obj = true; obj_name = name; }
            else free_rows.insert(name);
            continue;
        } else if (row_type == "L") op = OP_LE;
        else if (row_type == "G") op = OP_GE;
        else if (row_type == "E") op = OP_EQ;
        else throw err("unexpected row type " + row_type);
        if (constr_name2idx.count(name)) throw err("row " + name + " already declared");
        constr_name2idx[name] = constraints.size();
        constraints.push_back({{}, {}, op, 0.0, 0.0});
    }
    if (!have_obj) throw err("objective function name not declared");

    struct VariableDef {
        bool has_min = false, has_max = false;
        double min = 0, max = 0, obj_coeff = 0;
    };
    std::vector<VariableDef> var_defs;
    {  // mps.rs:131-176
        if (cur != "COLUMNS") throw err("expected COLUMNS section");
        usize cur_var = 0;
        std::string cur_name;
        VariableDef cur_def;
        for (;;) {
            to_next();
            if (!starts_space()) break;
            auto t = tokens_of(cur);
            const std::string& name = need(t, 0);
            if (name != cur_name) {
                if (out.variables.count(name)) throw err("variable " + name + " already declared");
                if (!cur_name.empty()) {
                    out.variables[cur_name] = cur_var;
                    out.var_names.push_back(cur_name);
                    var_defs.push_back(cur_def);
                    cur_def = VariableDef();
                    cur_var += 1;
                }
                cur_name = name;
            }
            for (auto& kv : kv_pairs(t, 1)) {
                if (kv.first == obj_name) cur_def.obj_coeff = kv.second;
                else if (constr_name2idx.count(kv.first)) {
                    ConstraintDef& c = constraints[constr_name2idx[kv.first]];
                    c.vars.push_back(cur_var);
                    c.coeffs.push_back(kv.second);
                } else if (!free_rows.count(kv.first)) throw err("unknown constraint: " + kv.first);
            }
        }
        if (!cur_name.empty()) {
            out.variables[cur_name] = cur_var;
            out.var_names.push_back(cur_name);
            var_defs.push_back(cur_def);
        }
    }
    {  // mps.rs:178-210
        if (cur != "RHS") throw err("expected RHS section");
        bool have_vec = false;
        std::string vec;
        for (;;) {
            to_next();
            if (!starts_space()) break;
            auto t = tokens_of(cur);
            const std::string& vn = need(t, 0);
            if (!have_vec) { have_vec = true; vec = vn; }
            else if (vec != vn) continue;
            for (auto& kv : kv_pairs(t, 1)) {
                if (kv.first == obj_name) throw err("setting objective in RHS section is not supported");
                else if (constr_name2idx.count(kv.first)) constraints[constr_name2idx[kv.first]].rhs = kv.second;
                else throw err("unknown constraint: " + kv.first);
            }
        }
    }
    if (cur == "RANGES") {  // mps.rs:212-238
        bool have_vec = false;
        std::string vec;
        for (;;) {
            to_next();
            if (!starts_space()) break;
            auto t = tokens_of(cur);
            const std::string& vn = need(t, 0);
            if (!have_vec) { have_vec = true; vec = vn; }
            else if (vec != vn) continue;
            for (auto& kv : kv_pairs(t, 1)) {
                if (constr_name2idx.count(kv.first)) constraints[constr_name2idx[kv.first]].range = kv.second;
                else throw err("unknown constraint: " + kv.first);
            }
        }
    }
    if (cur == "BOUNDS") {  // mps.rs:240-287
        bool have_vec = false;
        std::string vec;
        for (;;) {
            to_next();
            if (!starts_space()) break;
            auto t = tokens_of(cur);
            const std::string& bt = need(t, 0);
            const std::string& vn = need(t, 1);
            if (!have_vec) { have_vec = true; vec = vn; }
            else if (vec != vn) continue;
            const std::string& var_name = need(t, 2);
            if (!out.variables.count(var_name)) throw err("unknown variable: " + var_name);
            VariableDef& d = var_defs[out.variables[var_name]];
            if (bt == "FR") {
                d.has_min = d.has_max = true;
                d.min = -INF;
                d.max = INF;
                continue;
            }
            double val = parse_f64(need(t, 3));
            if (bt == "LO") { d.has_min = true; d.min = val; }
            else if (bt == "UP") { d.has_max = true; d.max = val; }
            else if (bt == "FX") { d.has_min = d.has_max = true; d.min = d.max = val; }
            else throw err("bound type " + bt + " is not supported");
        }
    }
    if (cur != "ENDATA") throw err("expected ENDATA section");

    out.problem.direction = direction;  // mps.rs:293-304
    for (const VariableDef& d : var_defs) {
        double mn, mx;
        if (d.has_min && d.has_max) { mn = d.min; mx = d.max; }
        else if (d.has_min) { mn = d.min; mx = INF; }
        else if (d.has_max && d.max < 0.0) { mn = -INF; mx = d.max; }
        else if (d.has_max) { mn = 0.0; mx = d.max; }
        else { mn = 0.0; mx = INF; }
        out.problem.add_var(d.obj_coeff, mn, mx);
    }
    for (ConstraintDef& c : constraints) {  // mps.rs:306-321
        if (c.range == 0.0) {
            out.problem.add_constraint(c.vars, c.coeffs, c.op, c.rhs);
        } else {
            double mn, mx;
            if (c.op == OP_GE) { mn = c.rhs; mx = c.rhs + std::fabs(c.range); }
            else if (c.op == OP_LE) { mn = c.rhs - std::fabs(c.range); mx = c.rhs; }
            else if (c.range > 0.0) { mn = c.rhs; mx = c.rhs + c.range; }
            else { mn = c.rhs + c.range; mx = c.rhs; }
            out.problem.add_constraint(c.vars, c.coeffs, OP_GE, mn);
            out.problem.add_constraint(c.vars, c.coeffs, OP_LE, mx);
        }
    }
    return out;
}

// ===================================================================== C ABI
// Status: 0 OK, 1 Infeasible, 2 Unbounded, -1 panic (message via orc_last_error).
static thread_local std::string g_last_error;

template <class F>
static int guarded(F f) {
    try {
        f();
        return 0;
    } catch (LpFail& e) {
        return (int)e.e;
    } catch (std::exception& e) {
        g_last_error = e.what();
        return -1;
    } catch (SingularMatrix&) {
        g_last_error = "SingularMatrix";
        return -2;
    }
}

extern "C" {

const char* orc_last_error() { return g_last_error.c_str(); }

// ---- Problem / Solution (lib.rs)
Problem* orc_problem_new(int direction) {
    Problem* p = new Problem();
    p->direction = direction;
    return p;
}
Problem* orc_problem_clone(const Problem* p) { return new Problem(*p); }
void orc_problem_free(Problem* p) { delete p; }
uint64_t orc_problem_add_var(Problem* p, double obj, double mn, double mx) { return p->add_var(obj, mn, mx); }
uint64_t orc_problem_num_vars(const Problem* p) { return p->obj_coeffs.size(); }
int orc_problem_add_constraint(Problem* p, const uint32_t* idx, const double* coef, uint64_t k, int op, double rhs) {
    return guarded([&] {
        std::vector<usize> i(idx, idx + k);
        std::vector<double> v(coef, coef + k);
        p->add_constraint(i, v, (ComparisonOp)op, rhs);
    });
}
// budget < 0: solve to optimality. budget >= 0: stop after that many simplex iterations
// (status 0 with orc_solution_budget_exhausted()==1 if the budget ran out first).
int orc_problem_solve_ex(const Problem* p, Solution** out, int64_t budget, int trace) {
    Solution* s = new Solution();
    int st = guarded([&] {
        s->direction = p->direction;
        s->num_vars = p->obj_coeffs.size();
        s->solver.pivot_budget = budget;
        s->solver.trace = trace != 0;
        s->solver.try_new(p->obj_coeffs, p->var_mins, p->var_maxs, p->constraints);  // lib.rs:292-297
        s->solver.initial_solve();                                                  // lib.rs:298
    });
    if (st != 0) {
        delete s;
        *out = nullptr;
    } else {
        *out = s;
    }
    return st;
}
int orc_problem_solve(const Problem* p, Solution** out) { return orc_problem_solve_ex(p, out, -1, 0); }
// Continue a budget-limited solve for `budget` more iterations.
int orc_solution_continue(Solution* s, int64_t budget) {
    return guarded([&] {
        s->solver.pivot_budget = budget;
        s->solver.budget_exhausted = false;
        s->solver.initial_solve();
    });
}
int orc_solution_budget_exhausted(const Solution* s) { return s->solver.budget_exhausted ? 1 : 0; }
void orc_solution_set_capture(Solution* s, int on) { s->solver.capture = on != 0; }

Solution* orc_solution_clone(const Solution* s) { return new Solution(*s); }
void orc_solution_free(Solution* s) { delete s; }
double orc_solution_objective(const Solution* s) { return s->objective(); }
uint64_t orc_solution_num_vars(const Solution* s) { return s->num_vars; }
int orc_solution_var_value(const Solution* s, uint64_t var, double* out) {
    return guarded([&] {
        if (var >= s->num_vars) throw OraclePanic("var out of range (lib.rs:345)");
        *out = s->solver.get_value(var);
    });
}
// Mutators mirror the consume-on-error semantics (lib.rs:359, 385): on a non-zero status the
// solution is freed and *s is set to NULL.
static int consume_on_error(Solution** s, int st) {
    if (st != 0) {
        delete *s;
        *s = nullptr;
    }
    return st;
}
int orc_solution_add_constraint(Solution** s, const uint32_t* idx, const double* coef, uint64_t k, int op, double rhs) {
    return consume_on_error(s, guarded([&] {
        std::vector<usize> i(idx, idx + k);
        std::vector<double> v(coef, coef + k);
        (*s)->solver.pivot_budget = -1;
        (*s)->solver.add_constraint(CsVec::make((*s)->num_vars, i, v), (ComparisonOp)op, rhs);  // lib.rs:375-379
    }));
}
int orc_solution_fix_var(Solution** s, uint64_t var, double val) {
    return consume_on_error(s, guarded([&] {
        if (var >= (*s)->num_vars) throw OraclePanic("var out of range (lib.rs:391)");
        (*s)->solver.pivot_budget = -1;
        (*s)->solver.fix_var(var, val);
    }));
}
int orc_solution_unfix_var(Solution** s, uint64_t var, int* was_fixed) {
    return consume_on_error(s, guarded([&] {
        if (var >= (*s)->num_vars) throw OraclePanic("var out of range (lib.rs:400)");
        (*s)->solver.pivot_budget = -1;
        *was_fixed = (*s)->solver.unfix_var(var) ? 1 : 0;
    }));
}
int orc_solution_add_gomory_cut(Solution** s, uint64_t var) {
    return consume_on_error(s, guarded([&] {
        if (var >= (*s)->num_vars) throw OraclePanic("var out of range (lib.rs:420)");
        (*s)->solver.pivot_budget = -1;
        (*s)->solver.add_gomory_cut(var);
    }));
}

// ---- instrumentation
struct OrcStats {
    uint64_t pivots, bound_flips, refactors, primal_iters, dual_iters, ftran, btran;
    uint64_t ftran_lu_nnz, btran_lu_nnz, eta_nnz_applied, row_sweep_nnz;
    uint64_t lu_nnz, eta_nnz, eta_count, num_constraints, num_total_vars;
    double t_refactor;
};
void orc_solution_stats(const Solution* s, OrcStats* o) {
    const Counters& c = s->solver.cnt;
    o->pivots = c.pivots; o->bound_flips = c.bound_flips; o->refactors = c.refactors;
    o->primal_iters = c.primal_iters; o->dual_iters = c.dual_iters; o->ftran = c.ftran; o->btran = c.btran;
    o->ftran_lu_nnz = c.ftran_lu_nnz; o->btran_lu_nnz = c.btran_lu_nnz; o->eta_nnz_applied = c.eta_nnz_applied;
    o->row_sweep_nnz = c.row_sweep_nnz;
    o->lu_nnz = s->solver.basis_solver.lu_factors.nnz();
    o->eta_nnz = s->solver.basis_solver.eta_matrices.coeff_cols.nnz();
    o->eta_count = s->solver.basis_solver.eta_matrices.len();
    o->num_constraints = s->solver.num_constraints();
    o->num_total_vars = s->solver.num_total_vars();
    o->t_refactor = c.t_refactor;
}
uint64_t orc_solution_trace_len(const Solution* s) { return s->solver.trace_log.size(); }
void orc_solution_trace_get(const Solution* s, uint64_t i, int32_t* phase, int64_t* col, int64_t* row,
                            int64_t* entering_var, int64_t* leaving_var, double* pivot_coeff, double* obj_after) {
    const PivotRecord& r = s->solver.trace_log[i];
    *phase = r.phase; *col = r.col; *row = r.row; *entering_var = r.entering_var; *leaving_var = r.leaving_var;
    *pivot_coeff = r.pivot_coeff; *obj_after = r.obj_after;
}

// White-box state access for the solver.rs:1391-1479 KATs. `what` selects the array; returns its
// length; copies min(len, cap) doubles (integers are converted) into out when out != NULL.
uint64_t orc_solution_state(const Solution* s, const char* what, double* out, uint64_t cap) {
    const Solver& v = s->solver;
    std::vector<double> tmp;
    std::string w(what);
    auto from_usize = [&](const std::vector<usize>& a) { tmp.assign(a.begin(), a.end()); };
    if (w == "orig_obj_coeffs") tmp = v.orig_obj_coeffs;
    else if (w == "orig_var_mins") tmp = v.orig_var_mins;
    else if (w == "orig_var_maxs") tmp = v.orig_var_maxs;
    else if (w == "orig_rhs") tmp = v.orig_rhs;
    else if (w == "basic_vars") from_usize(v.basic_vars);
    else if (w == "basic_var_vals") tmp = v.basic_var_vals;
    else if (w == "basic_var_mins") tmp = v.basic_var_mins;
    else if (w == "basic_var_maxs") tmp = v.basic_var_maxs;
    else if (w == "dual_edge_sq_norms") tmp = v.dual_edge_sq_norms;
    else if (w == "nb_vars") from_usize(v.nb_vars);
    else if (w == "nb_var_obj_coeffs") tmp = v.nb_var_obj_coeffs;
    else if (w == "nb_var_vals") tmp = v.nb_var_vals;
    else if (w == "primal_edge_sq_norms") tmp = v.primal_edge_sq_norms;
    else if (w == "cur_obj_val") tmp = {v.cur_obj_val};
    else if (w == "flags") tmp = {(double)v.is_primal_feasible, (double)v.is_dual_feasible,
                                   (double)v.enable_primal_steepest_edge, (double)v.enable_dual_steepest_edge};
    else if (w == "csr_indptr") from_usize(v.orig_constraints.indptr);
    else if (w == "csr_indices") from_usize(v.orig_constraints.indices);
    else if (w == "csr_data") tmp = v.orig_constraints.data;
    else if (w == "csc_indptr") from_usize(v.orig_constraints_csc.indptr);
    else if (w == "csc_indices") from_usize(v.orig_constraints_csc.indices);
    else if (w == "csc_data") tmp = v.orig_constraints_csc.data;
    else if (w == "col_coeffs") {  // alpha_q of the last pivot, dense by basic position (solver.rs:54)
        tmp.assign(v.basic_vars.size(), 0.0);
        for (usize p = 0; p < v.col_coeffs.len(); ++p) tmp[v.col_coeffs.indices[p]] = v.col_coeffs.values[p];
    } else if (w == "row_coeffs") tmp = v.row_coeffs.values;          // alpha_r, dense by non-basic position (solver.rs:57)
    else if (w == "sq_norms_update_helper") tmp = v.sq_norms_update_helper;
    else if (w == "inv_basis_row_coeffs") tmp = v.dbg_rho;             // rho_r by row (capture mode)
    else if (w == "dbg_v") tmp = v.dbg_v;                               // B^-T alpha_q by row (capture mode)
    else if (w == "dbg_tau") tmp = v.dbg_tau;                           // B^-1 rho_r by position (capture mode)
    else if (w == "nb_at_min") { for (auto& st : v.nb_var_states) tmp.push_back(st.at_min); }
    else if (w == "nb_at_max") { for (auto& st : v.nb_var_states) tmp.push_back(st.at_max); }
    else return (uint64_t)-1;
    if (out) for (usize i = 0; i < tmp.size() && i < cap; ++i) out[i] = tmp[i];
    return tmp.size();
}
// Build a Solver without solving (Solver::try_new only) for the `initialize` KAT.
int orc_problem_try_new(const Problem* p, Solution** out) {
    Solution* s = new Solution();
    int st = guarded([&] {
        s->direction = p->direction;
        s->num_vars = p->obj_coeffs.size();
        s->solver.try_new(p->obj_coeffs, p->var_mins, p->var_maxs, p->constraints);
    });
    if (st != 0) { delete s; *out = nullptr; } else *out = s;
    return st;
}

// ---- LU level (lu.rs KATs): factor a size x size matrix given in CSC.
struct OrcLU {
    LUFactors lu, lut;
    ScratchSpace scratch;
    usize size;
};
int orc_lu_factorize(uint64_t size, const uint64_t* indptr, const uint64_t* rows, const double* vals,
                     double stability, OrcLU** out) {
    OrcLU* h = new OrcLU();
    h->size = size;
    h->scratch = ScratchSpace(size);
    std::vector<usize> r(rows, rows + indptr[size]);
    int st = guarded([&] {
        GetCol gc = [&](usize c) { return ColView{r.data() + indptr[c], vals + indptr[c], (usize)(indptr[c + 1] - indptr[c])}; };
        h->lu = lu_factorize(size, gc, stability, h->scratch);
        h->lut = h->lu.transpose();
    });
    if (st != 0) { delete h; *out = nullptr; } else *out = h;
    return st;
}
void orc_lu_free(OrcLU* h) { delete h; }
uint64_t orc_lu_nnz(const OrcLU* h) { return h->lu.nnz(); }
// which: 0 L nondiag, 1 U nondiag (CSC of the factor; transp selects lu_transp). Returns nnz.
uint64_t orc_lu_get_factor(const OrcLU* h, int transp, int which, uint64_t* indptr, uint64_t* rows, double* vals, double* diag) {
    const LUFactors& f = transp ? h->lut : h->lu;
    const TriangleMat& t = which == 0 ? f.lower : f.upper;
    if (indptr) for (usize i = 0; i <= t.cols(); ++i) indptr[i] = t.nondiag.indptr[i];
    if (rows) for (usize i = 0; i < t.nondiag.nnz(); ++i) rows[i] = t.nondiag.indices[i];
    if (vals) for (usize i = 0; i < t.nondiag.nnz(); ++i) vals[i] = t.nondiag.data[i];
    if (diag) for (usize i = 0; i < t.cols(); ++i) diag[i] = t.has_diag ? t.diag[i] : 1.0;
    return t.nondiag.nnz();
}
int orc_lu_has_diag(const OrcLU* h, int transp, int which) {
    const LUFactors& f = transp ? h->lut : h->lu;
    return (which == 0 ? f.lower : f.upper).has_diag ? 1 : 0;
}
void orc_lu_get_perms(const OrcLU* h, int transp, uint64_t* row_new2orig, uint64_t* col_new2orig,
                      uint64_t* row_orig2new, uint64_t* col_orig2new) {
    const LUFactors& f = transp ? h->lut : h->lu;
    for (usize i = 0; i < h->size; ++i) {
        if (row_new2orig) row_new2orig[i] = f.row_perm.new2orig[i];
        if (col_new2orig) col_new2orig[i] = f.col_perm.new2orig[i];
        if (row_orig2new) row_orig2new[i] = f.row_perm.orig2new[i];
        if (col_orig2new) col_orig2new[i] = f.col_perm.orig2new[i];
    }
}
void orc_lu_solve_dense(OrcLU* h, int transp, double* rhs) {
    (transp ? h->lut : h->lu).solve_dense(rhs, h->size, h->scratch);
}
// Sparse solve (lu.rs:79-106). In: k (idx,val) pairs. Out: nonzero list in ScatteredVec order
// (out_idx/out_val sized >= size); returns the number of marked entries.
uint64_t orc_lu_solve_sparse(OrcLU* h, int transp, const uint64_t* idx, const double* val, uint64_t k,
                             uint64_t* out_idx, double* out_val) {
    ScatteredVec rhs = ScatteredVec::empty(h->size);
    std::vector<usize> i(idx, idx + k);
    rhs.set(i.data(), val, k);
    (transp ? h->lut : h->lu).solve(rhs, h->scratch);
    for (usize p = 0; p < rhs.nonzero.size(); ++p) {
        out_idx[p] = rhs.nonzero[p];
        out_val[p] = rhs.values[rhs.nonzero[p]];
    }
    return rhs.nonzero.size();
}

// ---- sparse.rs:230-269 transpose KAT
void orc_sparse_transpose(uint64_t n_rows, uint64_t n_cols, const uint64_t* indptr, const uint64_t* rows, const double* vals,
                          uint64_t* out_indptr, uint64_t* out_indices, double* out_data) {
    SparseMat m(n_rows);
    for (usize c = 0; c < n_cols; ++c) {
        for (usize p = indptr[c]; p < indptr[c + 1]; ++p) m.push(rows[p], vals[p]);
        m.seal_column();
    }
    SparseMat t = m.transpose();
    for (usize i = 0; i < t.indptr.size(); ++i) out_indptr[i] = t.indptr[i];
    for (usize i = 0; i < t.nnz(); ++i) {
        out_indices[i] = t.indices[i];
        out_data[i] = t.data[i];
    }
}

// ---- MPS (mps.rs)
struct OrcMps {
    MpsFile f;
};
int orc_mps_parse(const char* text, uint64_t len, int direction, OrcMps** out) {
    OrcMps* h = new OrcMps();
    int st = 0;
    try {
        h->f = mps_parse(std::string(text, len), direction);
    } catch (std::exception& e) {
        g_last_error = e.what();
        st = -1;
    }
    if (st != 0) { delete h; *out = nullptr; } else *out = h;
    return st;
}
void orc_mps_free(OrcMps* h) { delete h; }
const char* orc_mps_name(const OrcMps* h) { return h->f.problem_name.c_str(); }
uint64_t orc_mps_num_vars(const OrcMps* h) { return h->f.var_names.size(); }
const char* orc_mps_var_name(const OrcMps* h, uint64_t i) { return h->f.var_names[i].c_str(); }
int64_t orc_mps_var_index(const OrcMps* h, const char* name) {
    auto it = h->f.variables.find(name);
    return it == h->f.variables.end() ? -1 : (int64_t)it->second;
}
Problem* orc_mps_problem(const OrcMps* h) { return new Problem(h->f.problem); }

// Raw problem data access (used by tests to feed the identical problem to the product).
uint64_t orc_problem_num_constraints(const Problem* p) { return p->constraints.size(); }
void orc_problem_var(const Problem* p, uint64_t v, double* obj, double* mn, double* mx) {
    *obj = p->direction == 1 ? -p->obj_coeffs[v] : p->obj_coeffs[v];
    *mn = p->var_mins[v];
    *mx = p->var_maxs[v];
}
uint64_t orc_problem_constraint(const Problem* p, uint64_t c, uint32_t* idx, double* coef, uint64_t cap, int* op, double* rhs) {
    const Constraint& k = p->constraints[c];
    *op = (int)k.op;
    *rhs = k.rhs;
    for (usize i = 0; i < k.coeffs.indices.size() && i < cap; ++i) {
        idx[i] = (uint32_t)k.coeffs.indices[i];
        coef[i] = k.coeffs.data[i];
    }
    return k.coeffs.indices.size();
}

}  // extern "C"
