// engine.hip — host orchestration of the device-resident simplex.  Control flow mirrors the
// reference's Solver (solver.rs:108-1241); every per-pivot vector stays in HBM and every hot loop
// is a kernel from kernels.hip.  A simplex iteration is a fixed kernel sequence driven entirely by
// device-resident state (Ctl), captured once into a hipGraph and replayed in batches; the host
// reads back one small record per pivot, once per batch.
#include "engine.h"


#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mlp {

static const double INF = std::numeric_limits<double>::infinity();
static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ ProblemData (lib.rs:215-283)
int ProblemData::add_var(double c, double mn, double mx) {
    int v = (int)obj.size();
    obj.push_back(direction == 1 ? -c : c);  // lib.rs:235-238
    lo.push_back(mn);
    hi.push_back(mx);
    return v;
}
static Constraint make_constraint(const uint32_t* vars, const double* coeffs, uint64_t k, int op, double rhs, size_t dim) {
    // sprs CsVec::new behaviour relied on at lib.rs:279: sort by index, reject duplicates / out of range
    std::vector<size_t> ord(k);
    for (size_t i = 0; i < k; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return vars[a] < vars[b]; });
    Constraint c;
    c.op = op;
    c.rhs = rhs;
    c.idx.resize(k);
    c.val.resize(k);
    for (size_t i = 0; i < k; ++i) {
        c.idx[i] = (int)vars[ord[i]];
        c.val[i] = coeffs[ord[i]];
    }
    for (size_t i = 0; i + 1 < k; ++i)
        if (c.idx[i] == c.idx[i + 1]) throw MlpError(-1, "variable added more than once to a constraint (lib.rs:247-249)");
    if (k && (size_t)c.idx[k - 1] >= dim) throw MlpError(-1, "variable index out of range");
    return c;
}
void ProblemData::add_constraint(const uint32_t* vars, const double* coeffs, uint64_t k, int op, double rhs) {
    if (op < 0 || op > 2) throw MlpError(-1, "bad comparison op");
    cons.push_back(make_constraint(vars, coeffs, k, op, rhs, obj.size()));
}

// ------------------------------------------------------------------ Engine basics
Engine::Engine() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw MlpError(-4, "no HIP device visible: the simplex hot path has no CPU fallback");
    acquire_runtime();
    std::memset(h_ctl, 0, sizeof(Ctl));
    std::memset(&hview, 0, sizeof(hview));
    const char* ng = std::getenv("MLP_NO_GRAPH");
    use_graph = !(ng && ng[0] == '1');
    const char* rt = std::getenv("MLP_REFRESH_TOL");
    if (rt) refresh_tol = std::atof(rt);
    const char* lr = std::getenv("MLP_LOWRANK");
    if (lr) lr_force = std::max(0, std::min(LR_MAX, std::atoi(lr)));
    const char* lp = std::getenv("MLP_LDPAD");
    if (lp) ld_pad = std::max(0, std::min(4096, std::atoi(lp))) & ~1;
    const char* bt = std::getenv("MLP_BIGTILE");
    force_big_tiles = bt && std::atoi(bt) != 0;
    const char* dm = std::getenv("MLP_DETERMINISTIC");
    if (dm) det_mode = std::atoi(dm) != 0 ? 1 : 0;
    const char* bd = std::getenv("MLP_BANDED");
    if (bd) banded_mode = std::atoi(bd) != 0 ? 1 : 0;
    const char* fr = std::getenv("MLP_FINAL_REFRESH");
    if (fr) final_refresh_pivots = std::atol(fr);
    if (const char* sp = std::getenv("MLP_SWEEP_PACKED")) use_pack = sp[0] != '0';
    if (const char* of = std::getenv("MLP_ORDER_FROM")) order_from = (uint64_t)std::atoll(of);    // (tests: locality order from pivot 0,
    if (const char* oe = std::getenv("MLP_ORDER_EVERY")) order_every = (uint64_t)std::atoll(oe);  //  rebuilt every few pivots)
    const char* lz = std::getenv("MLP_LAZY_DSE");
    lazy_dse = !(lz && lz[0] == '0');
    ratio_two = std::getenv("MLP_RATIO_TWO_KERNELS") != nullptr;
    if (const char* hy = std::getenv("MLP_HYPER")) hyper_mode = std::atoi(hy) > 0 ? 1 : 0;  // 1: whenever the kernel applies, 0: never
    if (const char* phk = std::getenv("MLP_PRIMAL_HEAD_K")) ph_kmax = std::max(0, std::atoi(phk));
    if (const char* sbk = std::getenv("MLP_SMALL_BASIS_K")) sb_kmax = std::max(0, std::min(256, std::atoi(sbk)));
    if (const char* sk = std::getenv("MLP_STR_K")) str_kmax = std::atoi(sk);  // sparse tableau row up to this nucleus size (0: never)
    if (const char* hh = std::getenv("MLP_HYPER_HEAVY")) hyper_heavy = std::atol(hh);        // work bound per iteration (tests force bail-outs)
    if (const char* rs = std::getenv("MLP_RATIO_SPIN_LIMIT")) ratio_spin_limit = std::atoll(rs);
    if (const char* sb = std::getenv("MLP_STREAM_BALANCED")) sw_balanced = std::atoi(sb);
    if (const char* fm = std::getenv("MLP_FACTOR")) fac_mode = std::atoi(fm) > 0 ? 1 : 0;
    if (const char* fj = std::getenv("MLP_FACTOR_J")) {  // a fixed period; default: 48, and 64 while a refactorisation is expensive (fac_refactor)
        fac_J_ = fac_period_ = std::max(1, std::min(64, std::atoi(fj)));
        fac_period_auto_ = false;
    }
    if (const char* ff = std::getenv("MLP_FACTOR_FROM")) fac_auto_cap_ = std::max(256, std::atoi(ff));
    if (const char* fb = std::getenv("MLP_FACTOR_BUMP")) {  // (lowering the bump limit lowers it for both carriers of the bump)
        const int want = std::atoi(fb);
        fac_bump_max_ = std::max(0, std::min(FAC_BMAX, want));
        if (want < FAC_BMAX) fac_sb_max_ = std::min(fac_sb_max_, fac_bump_max_);  // (a value at or above the dense carrier's capacity leaves the sparse carrier's limit alone)
    }
    if (const char* fb = std::getenv("MLP_FACTOR_SB")) fac_sb_max_ = std::max(0, std::min(FAC_SB_MAX, std::atoi(fb)));
    if (const char* fb = std::getenv("MLP_FACTOR_SB_FROM")) fac_sb_from_ = std::max(1, std::atoi(fb));
    if (const char* fp = std::getenv("MLP_FPULL")) fpull_on_ = fp[0] != '0';
    const char* nws = std::getenv("MLP_NO_WSHARD");
    no_wshard = nws && std::atoi(nws) != 0;
    const char* bs = std::getenv("MLP_BATCH");
    if (bs) batch = std::max(1, std::min(RING, std::atoi(bs)));
    const char* gi = std::getenv("MLP_GRAPH_ITERS");
    if (gi) graph_iters = std::max(1, std::min(32, std::atoi(gi)));
}
// ------------------------------------------------------------------ per-Solution runtime objects
// Two streams, six events and the pinned Ctl mirror cost 4-5 ms to create and 3 ms to destroy; the
// TSP driver clones and drops a Solution per branch-and-bound node, so idle sets are recycled.
namespace {
struct RtBundle {
    int device = 0;
    hipStream_t st = nullptr, st2 = nullptr;
    hipEvent_t evFork[3] = {nullptr, nullptr, nullptr}, evJoin[3] = {nullptr, nullptr, nullptr};
    Ctl* h_ctl = nullptr;
};
struct RtPool {
    std::mutex mu;
    std::vector<RtBundle> idle;
};
RtPool& rt_pool() {
    static RtPool* p = new RtPool();
    return *p;
}
constexpr size_t kMaxIdleRuntimes = 64;
}  // namespace
void Engine::acquire_runtime() {
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    rt_device = dev;
    {
        RtPool& p = rt_pool();
        std::lock_guard<std::mutex> lk(p.mu);
        for (size_t i = 0; i < p.idle.size(); ++i)
            if (p.idle[i].device == dev) {
                RtBundle b = p.idle[i];
                p.idle.erase(p.idle.begin() + (long)i);
                st = b.st; st2 = b.st2; h_ctl = b.h_ctl;
                for (int j = 0; j < 3; ++j) { evFork[j] = b.evFork[j]; evJoin[j] = b.evJoin[j]; }
                return;
            }
    }
    try {
        HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        for (int i = 0; i < 3; ++i) {
            HIPCHECK(hipEventCreateWithFlags(&evFork[i], hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&evJoin[i], hipEventDisableTiming));
        }
        HIPCHECK(hipHostMalloc((void**)&h_ctl, sizeof(Ctl), hipHostMallocDefault));
    } catch (...) {
        release_runtime();  // destroys what was created so far (the destructor does not run for a throwing constructor)
        throw;
    }
}
void Engine::release_runtime() {  // both streams are idle (synchronised by the destructor)
    RtBundle b;
    b.device = rt_device; b.st = st; b.st2 = st2; b.h_ctl = h_ctl;
    for (int j = 0; j < 3; ++j) { b.evFork[j] = evFork[j]; b.evJoin[j] = evJoin[j]; }
    st = st2 = nullptr; h_ctl = nullptr;
    if (!b.st || !b.st2 || !b.h_ctl) {  // partially constructed: destroy what exists
        if (b.h_ctl) (void)hipHostFree(b.h_ctl);
        for (int j = 0; j < 3; ++j) {
            if (b.evFork[j]) (void)hipEventDestroy(b.evFork[j]);
            if (b.evJoin[j]) (void)hipEventDestroy(b.evJoin[j]);
        }
        if (b.st2) (void)hipStreamDestroy(b.st2);
        if (b.st) (void)hipStreamDestroy(b.st);
        return;
    }
    RtPool& p = rt_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.idle.size() < kMaxIdleRuntimes) {
        p.idle.push_back(b);
        return;
    }
    (void)hipHostFree(b.h_ctl);
    for (int j = 0; j < 3; ++j) {
        (void)hipEventDestroy(b.evFork[j]);
        (void)hipEventDestroy(b.evJoin[j]);
    }
    (void)hipStreamDestroy(b.st2);
    (void)hipStreamDestroy(b.st);
}

// ------------------------------------------------------------------ device block cache
namespace {
struct PoolState {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void*>> free_blocks;  // by (device, block size)
    size_t cached = 0;
};
PoolState& pool() {
    static PoolState* s = new PoolState();  // never destroyed: blocks may be returned during process exit
    return *s;
}
size_t round_block(size_t bytes) {
    size_t b = 256;
    while (b < bytes) b <<= 1;
    return b;
}
}  // namespace
// MLP_POOL_POISON=1 (debugging): every block handed out — recycled or fresh — is filled with 0xFF bytes first (NaN as a double,
// -1 as an int), so that a kernel reading memory nobody initialised shows up deterministically instead of depending on what
// the previous owner of the block left behind.
static bool pool_poison() {
    static const bool on = std::getenv("MLP_POOL_POISON") != nullptr && std::atoi(std::getenv("MLP_POOL_POISON")) != 0;
    return on;
}
void* DevPool::get(size_t bytes, size_t* got_bytes, int* device) {
    void* p = get_raw(bytes, got_bytes, device);
    if (pool_poison()) {
        (void)hipMemset(p, 0xFF, *got_bytes);
        (void)hipDeviceSynchronize();
    }
    return p;
}
void* DevPool::get_raw(size_t bytes, size_t* got_bytes, int* device) {
    if (bytes == 0) bytes = 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    *device = dev;
    if (bytes <= kMaxBlock) {
        const size_t b = round_block(bytes);
        {
            PoolState& s = pool();
            std::lock_guard<std::mutex> lk(s.mu);
            auto it = s.free_blocks.find(std::make_pair(dev, b));
            if (it != s.free_blocks.end() && !it->second.empty()) {
                void* p = it->second.back();
                it->second.pop_back();
                s.cached -= b;
                *got_bytes = b;
                return p;
            }
        }
        bytes = b;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {  // give the cache back to the runtime and retry once
        trim();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
        throw MlpError(-3, std::string("hipMalloc(") + std::to_string(bytes) + " bytes) failed: " + hipGetErrorString(e));
    *got_bytes = bytes;
    return p;
}
void DevPool::put(void* p, size_t bytes, int device) {
    if (!p) return;
    if (bytes <= kMaxBlock && bytes == round_block(bytes)) {
        PoolState& s = pool();
        std::lock_guard<std::mutex> lk(s.mu);
        if (s.cached + bytes <= kMaxCached) {
            s.free_blocks[std::make_pair(device, bytes)].push_back(p);
            s.cached += bytes;
            return;
        }
    }
    (void)hipFree(p);
}
void DevPool::trim() {
    PoolState& s = pool();
    std::lock_guard<std::mutex> lk(s.mu);
    for (auto& kv : s.free_blocks)
        for (void* p : kv.second) (void)hipFree(p);
    s.free_blocks.clear();
    s.cached = 0;
}

Engine::~Engine() {
    if (st) (void)hipStreamSynchronize(st);   // the buffers go back to the pool: nothing may still use them
    if (st2) (void)hipStreamSynchronize(st2);
    drop_graphs();
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    release_mailboxes();
    release_runtime();
}
void Engine::drop_graphs() {
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            for (int w = 0; w < 2; ++w) {
                if (gexec[a][b][w]) (void)hipGraphExecDestroy(gexec[a][b][w]);
                if (ggraph[a][b][w]) (void)hipGraphDestroy(ggraph[a][b][w]);
                gexec[a][b][w] = nullptr;
                ggraph[a][b][w] = nullptr;
            }
            graph_batches_in_geom = 0;
        }
}

static constexpr size_t kMailBoxBytesPerRank = sizeof(MailRec) * 2 * MAIL_KINDS;
Geom Engine::geom() const {
    Geom g;
    g.m = m_;
    g.n = num_vars;
    g.cap = cap_;
    double avg = num_vars > 0 ? (double)(h_rcol.size() - (size_t)m_) / (double)num_vars : 1.0;  // structural columns
    g.lanes = avg >= 40.0 ? 64 : (avg >= 6.0 ? 16 : 4);
    g.sweep_one = avg < 3.0 ? 1 : 0;  // (the 400 000-column transport solve 15.63 s against 15.82-15.95 with four lanes per two-entry column)
    g.sweep_variant = sweep_variant;
    g.big = (cap_ > 4096 || force_big_tiles) ? 1 : 0;
    const int lr = lr_force >= 0 ? lr_force : (cap_ >= 8192 ? 32 : 0);  // as in sync_view
    g.head_fused = (!fac_on_ && lr == 0 && max_col_nnz_ <= HEAD_LIST_CAP && max_row_nnz_ <= HEAD_LIST_CAP) ? 1 : 0;
    g.fac = fac_on_ ? 1 : 0;
    // in-kernel wait between the two Harris passes: never while the ranks of a sharded solve share a device (their grids
    // compete for the same CUs), never again after a wait has timed out once
    g.ratio_two = (ratio_two || (shard_world > 1 && ranks_share_device)) ? 1 : 0;
    g.str = (str_now && !stepping && !fac_on_) ? 1 : 0;
    g.sb = (g.str && sb_now) ? 1 : 0;
    g.ph = (g.str && ph_now) ? 1 : 0;
    g.fp = hview.fpk_on ? 1 : 0;
    return g;
}

// Band-major copy of A for the banded sweep: for band b, a CSC over ALL columns restricted to rows
// [b * BAND_ROWS, (b + 1) * BAND_ROWS) with 16-bit local row indices.  Columns hold ascending rows, so the
// entries of (column, band) are a contiguous run of the plain CSC.
bool Engine::use_banded() const {
    if (banded_mode == 0) return false;
    if (banded_mode == 1) return true;
    return m_ >= 4 * BAND_ROWS && h_rcol.size() >= ((size_t)1 << 22);
}
void Engine::ensure_banded() {
    if (!banded_dirty) return;
    // built on the device from the CSC (kernels.hip: k_band_count -> exclusive scan -> k_band_fill)
    const int nb = (m_ + BAND_ROWS - 1) / BAND_ROWS;
    const size_t np = (size_t)nb * (size_t)(N_ + 1);
    d_bptr.ensure(np, 0, st);
    d_scan_tmp.ensure(np / 4096 + 8, 0, st);
    launch_band_count(d_cptr.p, d_crow.p, N_, nb, d_bptr.p, st);
    launch_exclusive_scan(d_bptr.p, d_bptr.p, (long)np, d_scan_tmp.p, st);
    int total = 0;
    HIPCHECK(hipMemcpyAsync(&total, d_scan_tmp.p + (np + 4095) / 4096, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    d_brow.ensure((size_t)total + 8, 0, st);  // + 8: the kernel reads whole groups of 8 entries
    d_bval.ensure((size_t)total + 8, 0, st);
    band_total_ = (size_t)total;
    pack_built = false;  // (the packed non-basic copy is derived from this one)
    HIPCHECK(hipMemsetAsync(d_brow.p + total, 0, 8 * sizeof(unsigned short), st));
    HIPCHECK(hipMemsetAsync(d_bval.p + total, 0, 8 * sizeof(double), st));
    launch_band_fill(d_cptr.p, d_crow.p, d_cval.p, N_, nb, d_bptr.p, d_brow.p, d_bval.p, st);
    d_band_part.ensure((size_t)nb * (size_t)num_vars, 0, st);
    banded_dirty = false;
}

void Engine::refresh_nb_order(bool force) {
    if (!force && (!order_valid || lifetime_pivots - order_built_at < order_every)) return;
    std::vector<int> ord;
    ord.reserve((size_t)num_vars);
    for (int var = 0; var < N_; ++var)  // ascending variables = storage order of the band-major copy
        if (h_var_loc[var] < 0) ord.push_back(-1 - h_var_loc[var]);
    if ((int)ord.size() != num_vars) throw MlpError(-3, "refresh_nb_order: host mirror of var_loc is inconsistent");
    d_nb_order.ensure((size_t)num_vars, 0, st);
    HIPCHECK(hipMemcpyAsync(d_nb_order.p, ord.data(), sizeof(int) * (size_t)num_vars, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));  // `ord` is a local staging buffer
    order_built_at = lifetime_pivots;
    order_valid = true;
    // the packed copy of the current non-basic columns in that order (device: lengths -> exclusive scan -> copy)
    pack_built = false;
    if (use_pack && band_total_ > 0 && !banded_dirty) {
        const int nb = (m_ + BAND_ROWS - 1) / BAND_ROWS;
        const size_t np = (size_t)nb * ((size_t)num_vars + 1);
        try {
            d_pk_ptr.ensure(np, 0, st);
            {   // (the sweep reads whole groups of 8 entries: what lies beyond the packed total must be valid row indices and
                // finite values — zero a fresh allocation once; later refills leave entries of older copies there, which are)
                const unsigned short* r0 = d_pk_row.p;
                const double* v0 = d_pk_val.p;
                d_pk_row.ensure(band_total_ + 16, 0, st);
                d_pk_val.ensure(band_total_ + 16, 0, st);
                if (d_pk_row.p != r0) HIPCHECK(hipMemsetAsync(d_pk_row.p, 0, sizeof(unsigned short) * d_pk_row.cap, st));
                if (d_pk_val.p != v0) HIPCHECK(hipMemsetAsync(d_pk_val.p, 0, sizeof(double) * d_pk_val.cap, st));
            }
            d_pk_valid.ensure((size_t)num_vars, 0, st);
            d_scan_tmp.ensure(np / 4096 + 8, 0, st);
        } catch (MlpError&) {  // no room for the second copy: the sweep keeps the indirect path
            (void)hipGetLastError();
            use_pack = false;
            return;
        }
        DevView t{};
        t.m = m_; t.n = num_vars; t.nbands = nb;
        t.bptr = d_bptr.p; t.brow = d_brow.p; t.bval = d_bval.p;
        t.nb_vars = d_nb_vars.p; t.nb_order = d_nb_order.p;
        launch_pack_count(t, d_pk_ptr.p, st);
        launch_exclusive_scan(d_pk_ptr.p, d_pk_ptr.p, (long)np, d_scan_tmp.p, st);
        launch_pack_fill(t, d_pk_ptr.p, d_pk_row.p, d_pk_val.p, d_pk_valid.p, st);
        HIPCHECK(hipStreamSynchronize(st));
        pack_built = true;
    }
}

// Row-block offsets of every column for the blocked F push: colblk[var][b] = first CSC index of column
// var whose row is >= b * PB_ROWS (columns hold ascending rows), colblk[var][RB] = end of the column.
void Engine::ensure_colblk() {
    if (!colblk_dirty) return;
    const int rb = (m_ + PB_ROWS - 1) / PB_ROWS;
    d_colblk.ensure((size_t)N_ * (rb + 1), 0, st);
    launch_build_colblk(d_cptr.p, d_crow.p, N_, rb, d_colblk.p, st);  // on the device, from the CSC
    d_push_part.ensure((size_t)PB_CHUNKS * (size_t)m_, 0, st);
    colblk_dirty = false;
}

// Pulled F product (fpull.inc): wanted wherever the blocked push in its float-atomic form serves the unsharded delayed-update mode
bool Engine::fpk_wanted() const {
    const int lr = fac_on_ ? 0 : (lr_force >= 0 ? lr_force : (cap_ >= 8192 ? 32 : 0));
    // (sharded solves too: the pull is replicated on every rank like the rest of the FTRAN, and — a fixed summation order — it keeps the ranks
    // bit-identical replicas without the fixed-point limbs of the deterministic push)
    return fpull_on_ && !fac_on_ && !stepping && lr > 0 && enable_pse && d_rowinfo.p != nullptr;
}
// (Re)build the row-major packed copy of the nucleus columns from the CSR of A and the device's CURRENT maps, on the device, in one
// pass (~30 us on config 4: 10^7 entries read once); alpha_K by variable starts from zero.  Called between batches: at the first batch
// of every run_loop, and every fpk_every_ pivots to drop the entries of columns that have left the basis since.
void Engine::fpk_rebuild() {
    const size_t nnz = h_rcol.size();
    d_fpk_cnt.ensure((size_t)m_ + 8, 0, st);
    d_fpk_var.ensure(nnz + 8, 0, st);
    d_fpk_val.ensure(nnz + 8, 0, st);
    d_fpk_in.ensure((size_t)N_ + 8, 0, st);
    d_fpk_x.ensure((size_t)N_ + 8, 0, st);
    d_fpk_part.ensure(2 * (((size_t)m_ * 2 + 2048) / 32 + 8), 0, st);  // (one pair per block of k_fpull_p1: 32 items of m rows + cap <= m slots each)
    DevView t = *sync_view();
    t.fpk_cnt = d_fpk_cnt.p; t.fpk_var = d_fpk_var.p; t.fpk_val = d_fpk_val.p; t.fpk_in = d_fpk_in.p; t.fpk_x = d_fpk_x.p;
    HIPCHECK(hipMemsetAsync(d_fpk_in.p, 0, (size_t)N_, st));
    HIPCHECK(hipMemsetAsync(d_fpk_x.p, 0, sizeof(double) * (size_t)N_, st));
    launch_fpk_build(t, st);
    fpk_valid_ = true;
    fpk_built_at_ = lifetime_pivots;
    fpk_builds_ += 1;
    if (!hview.fpk_on || hview.fpk_cnt != d_fpk_cnt.p || hview.fpk_var != d_fpk_var.p || hview.fpk_val != d_fpk_val.p ||
        hview.fpk_in != d_fpk_in.p || hview.fpk_x != d_fpk_x.p || hview.fpk_part != d_fpk_part.p) {
        view_dirty = true;
        sync_view();
    }
}

DevView* Engine::sync_view() {
    if (!view_dirty) return &hview;
    DevView old = hview;
    DevView& v = hview;
    v.m = m_; v.n = num_vars; v.ld = ld(); v.pad0 = 0;
    v.csc_ptr = d_cptr.p; v.csc_row = d_crow.p; v.csc_val = d_cval.p;
    v.csr_ptr = d_rptr.p; v.csr_col = d_rcol.p; v.csr_val = d_rval.p;
    v.var_lo = d_lo.p; v.var_hi = d_hi.p; v.obj_c = d_obj.p;
    v.var_loc = d_var_loc.p;
    v.basic_vars = d_basic_vars.p; v.xB = d_xB.p; v.loB = d_loB.p; v.hiB = d_hiB.p; v.beta = d_beta.p;
    v.nb_vars = d_nb_vars.p; v.d = d_d.p; v.xN = d_xN.p; v.gamma = d_gamma.p; v.nbflags = d_nbflags.p;
    v.kslot_of_pos = d_kslot_of_pos.p; v.srow_of_pos = d_srow_of_pos.p; v.sdiag_of_pos = d_sdiag_of_pos.p;
    v.kslot_of_row = d_kslot_of_row.p; v.pos_of_srow = d_pos_of_srow.p; v.rowinfo = d_rowinfo.p;
    // A sharded solve needs its ranks to be BIT-IDENTICAL replicas of the basic side (alpha_q, x_B, W, rho): the reduced costs
    // and weights live per column block, every rank updates its block with ITS rho and v but with the step rank 0 decided from
    // the entering column's owner — consistent only while all ranks hold the same numbers.  The F products through float
    // atomics (blocked push: LDS atomics; small nucleus: global atomics) are reproducible to rounding only; at config-4 size
    // that rounding-level difference between the ranks grows (each block's implicit dual vector is corrected by the OTHER
    // block's entering columns: an expansive coupled recurrence) until the reduced costs are off by 1e-1 relative and the
    // solve makes 5x less progress per pivot (found in round 3: DESIGN.md §6, profiles/r03g_sharded_long_run.log).  So a
    // sharded solve takes the deterministic pull (k_pull_F: fixed summation order), as small models do anyway.
    const bool shard_det = shard_world > 1;
    // Round 4: the blocked push has a DETERMINISTIC form (k_push_stage1_det: fixed-point limbs, integer LDS atomics — exact,
    // hence independent of the order in which they land), so a sharded solve no longer needs the pull for a nucleus beyond a
    // few hundred columns: it takes the blocked push from capacity 512 on (below that the marked pull touches few rows and
    // is cheaper than the push's fixed cost).  MLP_PB_DET=1 / 0 forces the deterministic form on / off for unsharded solves.
    static const int pb_det_env = std::getenv("MLP_PB_DET") ? std::atoi(std::getenv("MLP_PB_DET")) : -1;
    int hbits = 1;
    while (hbits < 31 && (1 << hbits) <= max_row_nnz_) hbits += 1;  // terms one accumulator can receive: < 2^hbits
    const bool det_fits = hbits <= 20;
    const bool pb_det = det_fits && (shard_det || force_det_push_ || pb_det_env == 1 || (pb_det_env < 0 && pb_det_default));
    // (small models take the pull sharded or not — MLP_DETERMINISTIC auto — so that their sharded and unsharded runs stay bit-identical)
    const bool pb_size = cap_ > 4096 || force_big_tiles ||
                         (shard_det && pb_det && cap_ >= 512 && det_mode != 1 && h_rcol.size() > ((size_t)1 << 21));
    v.pb_on = (pb_size && (!shard_det || pb_det)) ? 1 : 0;
    v.pb_det = (v.pb_on && pb_det) ? 1 : 0;
    v.pb_hbits = hbits;
    v.pb_amax = amax_;
    if (v.pb_on) ensure_colblk();
    v.colblk = v.pb_on ? d_colblk.p : nullptr;
    v.push_part = v.pb_on ? d_push_part.p : nullptr;
    v.pb_rb = (m_ + PB_ROWS - 1) / PB_ROWS;
    {   // pulled F product (fpull.inc): on while a valid packed copy exists for the mode it serves; the pointers stay in the view either way
        // (the update kernel keeps alpha_K-by-variable zero at the leaving variable whichever path ran the pivot)
        // (a view without it does not append the entering columns: the copy is stale from then on)
        if (!(fpk_wanted() && v.pb_on && (!v.pb_det || shard_world > 1))) fpk_valid_ = false;
        const bool fp = fpk_valid_ && d_fpk_cnt.p != nullptr;
        v.fpk_cnt = d_fpk_cnt.p; v.fpk_var = d_fpk_var.p; v.fpk_val = d_fpk_val.p; v.fpk_in = d_fpk_in.p;
        v.fpk_x = fp ? d_fpk_x.p : nullptr; v.fpk_part = d_fpk_part.p;
        v.fpk_on = fp ? 1 : 0; v.pad5 = 0;
    }
    v.det_pull = (!v.pb_on && (det_mode == 1 || shard_det || force_det_push_ || (det_mode < 0 && h_rcol.size() <= ((size_t)1 << 21)))) ? 1 : 0;
    if (v.det_pull) {  // (a pooled block is not zero: cleared whenever the allocation changes; the pull clears what it consumes)
        const unsigned char* before = d_fmark.p;
        d_fmark.ensure((size_t)m_ + 64, 0, st);
        if (d_fmark.p != before) HIPCHECK(hipMemsetAsync(d_fmark.p, 0, d_fmark.cap, st));
        v.fmark = d_fmark.p;
    } else {
        v.fmark = nullptr;
    }
    v.banded = use_banded() ? 1 : 0;
    if (v.banded) ensure_banded();
    v.bptr = v.banded ? d_bptr.p : nullptr;
    v.brow = v.banded ? d_brow.p : nullptr;
    v.bval = v.banded ? d_bval.p : nullptr;
    v.band_part = v.banded ? d_band_part.p : nullptr;
    v.nbands = (m_ + BAND_ROWS - 1) / BAND_ROWS;
    // (while the non-basic positions still hold their original variables the plain position order IS the locality
    // order and the indirection only costs: 96.9 vs 95.1 us per pivot in the benchmark window)
    if (v.banded && use_order && shard_world == 1 && (lifetime_pivots >= order_from || order_force)) {
        if (!order_valid || (use_pack && !pack_built)) refresh_nb_order(true);
        v.nb_order = d_nb_order.p;
    } else {
        v.nb_order = nullptr;
    }
    const bool pk = v.nb_order && use_pack && pack_built;
    v.pk_ptr = pk ? d_pk_ptr.p : nullptr; v.pk_row = pk ? d_pk_row.p : nullptr; v.pk_val = pk ? d_pk_val.p : nullptr;
    v.pk_valid = pk ? d_pk_valid.p : nullptr;
    v.pos_of_kslot = d_pos_of_kslot.p; v.row_of_kslot = d_row_of_kslot.p; v.W = d_W.p;
    v.U = d_U.p; v.V = d_V.p; v.Ut = d_Ut.p;
    // delayed-update period: 32 from capacity 8192 on (the fold's k^2 cost outgrows the O(k J) overheads)
    // (16 up to cap 16 384 in round 1: 304 vs 296 us per pivot at k = 10 000).  A period of 64 was measured in round 3 at
    // k = 20 500: the fused fold REPLACES that pivot's streaming pass, so its net cost is (1 338 - 479) / 32 = 27 us per pivot
    // and a period of 64 could save 13 of them at best; the 64-term kernel (occupancy 2) took 3.2 ms: 716 vs 695 us per pivot
    v.lrJ = fac_on_ ? 0 : (lr_force >= 0 ? lr_force : (cap_ >= 8192 ? 32 : 0));  // (compact factor: Ctl.nlow counts ITS rank-1 terms)
    v.alpha_q = d_work.p;
    v.tau = d_work.p + (size_t)m_;
    v.rv = reinterpret_cast<double2*>(d_work.p + 2 * (size_t)m_);
    v.alpha_r = d_alpha_r.p; v.helper = d_helper.p;
    v.aK = d_aK.p; v.rK = d_rK.p; v.tK = d_tK.p; v.tauK = d_tauK.p; v.vK = d_vK.p;
    v.klist_s = d_klist_s.p; v.klist_a = d_klist_a.p; v.blist_s = d_blist_s.p; v.blist_a = d_blist_a.p;
    v.part_tau = d_part_tau.p; v.part_v = d_part_v.p;
    v.red_key = d_red_key.p; v.red_key2 = d_red_key2.p; v.red_idx = d_red_idx.p; v.ticket = d_ticket.p;
    v.ctl = d_ctl.p;
    v.nb_rng = d_nb_rng.p;
    v.hy_stamp_n = d_hy_stamp.p;
    v.hy_stamp_p = d_hy_stamp.p ? d_hy_stamp.p + num_vars : nullptr;
    v.hy_var_slot = d_hy_stamp.p ? d_hy_stamp.p + num_vars + m_ : nullptr;
    v.hy_brng = d_hy_stamp.p ? reinterpret_cast<int2*>(d_hy_stamp.p + ((2 * ((size_t)num_vars + m_) + 1) & ~(size_t)1)) : nullptr;
    v.hy_score = d_hy_score.p;
    v.str_on = (str_now && !stepping && !fac_on_) ? 1 : 0;
    fac_fill_view(v);
    v.pad4 = 0;
    v.str_list = d_str_list.p;
    {   // list of alpha_r's non-zeros for the dual Harris test (one GPU; the CSC-pull tableau row builds it, kernels.hip ratio_dual_list)
        if (shard_world == 1) d_ar_list.ensure((size_t)num_vars + 64, 0, st);
        v.ar_list = shard_world == 1 ? d_ar_list.p : nullptr;
    }
    v.aq_list = d_str_list.p ? d_str_list.p + (((size_t)num_vars + 63) & ~(size_t)63) : nullptr;
    const bool live = shard_is_live();  // (deferred sharding: until it goes live the kernels see one rank that owns every column)
    // (a world of ONE — the rccl transport's self-test on a one-GPU box — keeps its mailbox: the handshake and the pump kernels go through it)
    v.rank = live ? shard_rank : 0; v.world = live ? shard_world : 1; v.mail = (live || shard_world == 1) ? d_mail : nullptr;
    for (int r = 0; r < MAX_WORLD; ++r) v.mail_peer[r] = reinterpret_cast<MailRec*>(peer_box[r]);
    v.mail_fanout = live ? mail_fanout : 0;
    {   // exchange buffers of the row-sharded streaming pass (peer transport, large-nucleus delayed-update mode only)
        const size_t mb = kMailBoxBytesPerRank * (size_t)std::max(shard_world, 1);
        v.wshard = (live && mail_fanout > 1 && own_box && v.lrJ > 0 && geom().big && stream_strips_enabled() &&
                    !no_wshard && (size_t)cap_ <= xb_cap_) ? 1 : 0;
        v.xbuf = own_box ? reinterpret_cast<double*>(static_cast<uint8_t*>(own_box) + mb) : nullptr;
        for (int r = 0; r < MAX_WORLD; ++r)
            v.xbuf_peer[r] = peer_box[r] ? reinterpret_cast<double*>(static_cast<uint8_t*>(peer_box[r]) + mb) : nullptr;
        v.xb_cap = (int)xb_cap_; v.pad3 = 0;
    }
    v.sw_nbal = 0; v.sw_pad = 0;
    if (sw_balanced && v.lrJ > 0 && geom().big && stream_strips_enabled())
        v.sw_nbal = sw_balanced > 1 ? sw_balanced : stream_coresident_blocks();
    v.nb_lo = live ? (int)((long)num_vars * shard_rank / shard_world) : 0;
    v.nb_hi = live ? (int)((long)num_vars * (shard_rank + 1) / shard_world) : num_vars;
    if (std::memcmp(&old, &hview, sizeof(DevView)) != 0) {
        HIPCHECK(hipStreamSynchronize(st));
        drop_graphs();  // kernel arguments (the view, by value) are baked into the captured graphs
        eager_iters_in_geom = 0;
    }
    view_dirty = false;
    return &hview;
}

void Engine::build_csc() {  // counting transpose of the CSR: ascending row index inside each column
    h_cptr.assign(N_ + 1, 0);
    for (int c : h_rcol) h_cptr[c + 1] += 1;
    for (int c = 0; c < N_; ++c) h_cptr[c + 1] += h_cptr[c];
    h_crow.resize(h_rcol.size());
    h_cval.resize(h_rcol.size());
    std::vector<int> next(h_cptr.begin(), h_cptr.end() - 1);
    for (int r = 0; r < m_; ++r)
        for (int p = h_rptr[r]; p < h_rptr[r + 1]; ++p) {
            int dst = next[h_rcol[p]]++;
            h_crow[dst] = r;
            h_cval[dst] = h_rval[p];
        }
    max_col_nnz_ = 0;
    max_row_nnz_ = 0;
    for (int r = 0; r < m_; ++r) max_row_nnz_ = std::max(max_row_nnz_, h_rptr[r + 1] - h_rptr[r]);
    amax_ = 0.0;
    for (double a : h_rval) amax_ = std::max(amax_, std::fabs(a));  // scale bound of the deterministic blocked push
    h_colnnz.resize(N_);
    h_single_row.assign(N_, -1);
    h_single_val.assign(N_, 0.0);
    for (int c = 0; c < N_; ++c) {
        h_colnnz[c] = h_cptr[c + 1] - h_cptr[c];
        max_col_nnz_ = std::max(max_col_nnz_, h_colnnz[c]);
        if (h_colnnz[c] == 1) {
            h_single_row[c] = h_crow[h_cptr[c]];
            h_single_val[c] = h_cval[h_cptr[c]];
        }
    }
}
void Engine::upload_matrix() {
    d_cptr.upload(h_cptr, st); d_crow.upload(h_crow, st); d_cval.upload(h_cval, st);
    d_rptr.upload(h_rptr, st); d_rcol.upload(h_rcol, st); d_rval.upload(h_rval, st);
    d_lo.upload(h_lo, st); d_hi.upload(h_hi, st); d_obj.upload(h_obj, st);
    colblk_dirty = true;
    banded_dirty = true;
    fpk_valid_ = false;  // (the packed copy of the nucleus columns is a copy of matrix entries)
    view_dirty = true;
}

void Engine::alloc_row_buffers(int m_new) {
    size_t keep = (size_t)m_;
    size_t mm = (size_t)m_new;
    d_basic_vars.ensure(mm, keep, st); d_xB.ensure(mm, keep, st); d_loB.ensure(mm, keep, st);
    d_hiB.ensure(mm, keep, st); d_beta.ensure(mm, keep, st);
    d_kslot_of_pos.ensure(mm, keep, st); d_srow_of_pos.ensure(mm, keep, st); d_sdiag_of_pos.ensure(mm, keep, st);
    d_kslot_of_row.ensure(mm, keep, st); d_pos_of_srow.ensure(mm, keep, st); d_rowinfo.ensure(mm, keep, st);
    d_work.ensure(5 * mm + 8, 0, st);
    d_klist_s.ensure(mm, 0, st); d_klist_a.ensure(mm, 0, st);
    d_blist_s.ensure(mm + num_vars + 64, 0, st); d_blist_a.ensure(mm + num_vars + 64, 0, st);
    d_var_loc.ensure((size_t)num_vars + mm, (size_t)num_vars + keep, st);
    view_dirty = true;
}

void Engine::ensure_nucleus_cap(int need) {
    if (fac_on_) return;       // compact factor: there is no nucleus inverse to grow
    if (need > m_) need = m_;  // the nucleus can never exceed the number of rows
    if (need <= cap_) return;
    // Selection of the second representation by the MEASURED shape of the basis: before the explicit inverse
    // grows past fac_auto_cap_ slots (8 k^2 bytes, O(k^2) traffic per pivot), peel the current basis; if the peel consumes
    // every column (no bump: the LU of this basis has no fill) the solve continues on the compact factor.  A basis that does
    // not peel (config 4: a random sparse nucleus fills to dense) keeps the explicit inverse; the attempt is repeated only
    // after the nucleus has doubled again.
    if (fac_mode < 0 && need > fac_auto_cap_ && fac_allowed_under_sharding() && !stepping && d_ctl.p && k_ > 0 &&
        (!fac_tried_ || cap_ >= 2 * (int)fac_tried_at_)) {
        fac_tried_ = true;
        fac_tried_at_ = (uint64_t)std::max(cap_, 1);
        if (fac_enter(32)) return;  // (automatic selection: only a basis that peels almost completely — a bump that grows would flip back)
    }
    flush_lowrank();  // pending rank-1 terms are folded before the buffers move
    HIPCHECK(hipStreamSynchronize(st));
    // Growth: doubling while the inverse is small; from 8 192 slots on (where 8 cap^2 bytes start to matter: 0.5 GB) by a
    // quarter, rounded up to 2 048 slots — a nucleus of 20 493 columns sits in a 24 576-slot array (4.8 GB) instead of a
    // 32 768-slot one (8.6 GB): at most 1.56x the bytes of the inverse instead of up to 4x (round-3 review, weak 10).  A growth
    // step costs one copy of the inverse (16 k^2 bytes: ~1.5 ms at k = 20 000) and a re-capture of the graphs.
    auto next_cap = [](long c) { return c < 8192 ? std::max(256L, c * 2) : ((c + c / 4 + 2047) / 2048) * 2048; };
    long ncap_l = next_cap((long)cap_);
    while (ncap_l < need) ncap_l = next_cap(ncap_l);
    // geometric growth, but never beyond the number of rows (rounded up to the column tile of the fused pass):
    // a 100 000-row model whose nucleus passes 65 536 slots gets a 100 352-slot W, not a 131 072-slot one
    const long m_cap = (((long)m_ + FW_TC - 1) / FW_TC) * FW_TC;
    if (ncap_l > m_cap && m_cap >= need) ncap_l = m_cap;
    const int ncap = (int)ncap_l;
    DevBuf<double> nW;
    const int old_ld = ld();
    const int nld = ncap + pad_for(ncap);
    try {
        nW.alloc_exact((size_t)ncap * nld);
    } catch (MlpError& e) {
        // the dense nucleus inverse is the memory wall of this design (8 k^2 bytes): report it as
        // such instead of a bare HIP error; the solution stays valid at its current capacity
        throw MlpError(-5, "nucleus inverse does not fit: " + std::to_string(ncap) + " x " + std::to_string(nld) +
                               " doubles (" + std::to_string(((size_t)ncap * nld * 8) >> 20) + " MiB) for a nucleus of " +
                               std::to_string(need) + " columns; " + e.what());
    }
    if (k_ > 0)
        HIPCHECK(hipMemcpy2DAsync(nW.p, (size_t)nld * sizeof(double), d_W.p, (size_t)old_ld * sizeof(double),
                                  (size_t)k_ * sizeof(double), (size_t)k_, hipMemcpyDeviceToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    d_W.release();
    d_W.p = nW.p; d_W.cap = nW.cap; d_W.bytes = nW.bytes; d_W.dev = nW.dev; nW.p = nullptr; nW.cap = 0; nW.bytes = 0;
    size_t keep = (size_t)k_;
    d_pos_of_kslot.ensure(ncap, keep, st); d_row_of_kslot.ensure(ncap, keep, st);
    d_aK.ensure(ncap, keep, st); d_rK.ensure(ncap, keep, st); d_tK.ensure(ncap, keep, st);
    d_tauK.ensure(ncap, keep, st); d_vK.ensure(ncap, keep, st);
    int nstripes = (ncap + FW_TR - 1) / FW_TR + 1, nchunks = (ncap + 512 - 1) / 512 + 1;  // 512: column chunks of k_stream_w
    d_part_v.ensure((size_t)nstripes * nld, 0, st);
    d_part_tau.ensure((size_t)nchunks * nld, 0, st);
    d_U.ensure((size_t)LR_MAX * nld, 0, st);
    d_V.ensure((size_t)LR_MAX * nld, 0, st);
    {   // slot-major copy of U for the fold (pending terms were folded above: it starts empty; stale entries beyond nlow are
        // multiplied by zero, so they must be finite: zero them once per allocation)
        const size_t before = d_Ut.cap;
        d_Ut.ensure((size_t)LR_MAX * nld, 0, st);
        if (d_Ut.cap != before) HIPCHECK(hipMemsetAsync(d_Ut.p, 0, sizeof(double) * d_Ut.cap, st));
    }
    cap_ = ncap;
    view_dirty = true;
}

void Engine::ensure_red() {
    // (k_fpull_p1 reduces over one block per 32 items of m rows + cap slots, cap <= m)
    size_t need = std::max<size_t>(1024, std::max((size_t)(std::max(m_, num_vars) + 255) / 256, ((size_t)m_ * 2 + 2048) / 32) + 8);
    d_red_key.ensure(need, 0, st); d_red_key2.ensure(need, 0, st); d_red_idx.ensure(need, 0, st);
    view_dirty = true;
}
// delayed-update mode: fold the pending rank-1 terms into W0 (host-requested, outside the pivot graph)
void Engine::flush_lowrank() {
    if (cap_ == 0 || !d_ctl.p) return;
    sync_view();
    if (!hview.lrJ) return;
    launch_fold_lowrank(hview, geom(), st);
    HIPCHECK(hipStreamSynchronize(st));
}
void Engine::pull_ctl() {
    HIPCHECK(hipGetLastError());  // a rejected kernel launch (bad configuration) must not pass silently
    HIPCHECK(hipMemcpyAsync(h_ctl, d_ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    k_ = h_ctl->k;
}
static void d2h_i(std::vector<int>& h, const DevBuf<int>& d, size_t n, hipStream_t st) {
    h.resize(n);
    if (n) HIPCHECK(hipMemcpyAsync(h.data(), d.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
}
void Engine::pull_maps() {
    HIPCHECK(hipStreamSynchronize(st));
    pull_ctl();
    size_t mm = m_;
    d2h_i(h_kslot_of_pos, d_kslot_of_pos, mm, st); d2h_i(h_srow_of_pos, d_srow_of_pos, mm, st);
    d2h_i(h_kslot_of_row, d_kslot_of_row, mm, st); d2h_i(h_pos_of_srow, d_pos_of_srow, mm, st);
    d2h_i(h_pos_of_kslot, d_pos_of_kslot, (size_t)cap_, st); d2h_i(h_row_of_kslot, d_row_of_kslot, (size_t)cap_, st);
    h_sdiag_of_pos.resize(mm);
    if (mm) HIPCHECK(hipMemcpyAsync(h_sdiag_of_pos.data(), d_sdiag_of_pos.p, mm * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
}
void Engine::push_maps() {
    d_kslot_of_pos.upload(h_kslot_of_pos, st); d_srow_of_pos.upload(h_srow_of_pos, st);
    d_sdiag_of_pos.upload(h_sdiag_of_pos, st);
    d_kslot_of_row.upload(h_kslot_of_row, st); d_pos_of_srow.upload(h_pos_of_srow, st);
    std::vector<RowInfo> ri(h_kslot_of_row.size());
    for (size_t i = 0; i < ri.size(); ++i) {
        if (h_kslot_of_row[i] >= 0) ri[i] = RowInfo{0.0, -1, h_kslot_of_row[i]};
        else ri[i] = RowInfo{h_sdiag_of_pos[h_pos_of_srow[i]], h_pos_of_srow[i], -1};
    }
    d_rowinfo.upload(ri, st);
    HIPCHECK(hipStreamSynchronize(st));  // `ri` is a local staging buffer
    h_pos_of_kslot.resize(cap_, -1);
    h_row_of_kslot.resize(cap_, -1);
    d_pos_of_kslot.upload(h_pos_of_kslot, st); d_row_of_kslot.upload(h_row_of_kslot, st);
    HIPCHECK(hipMemcpyAsync(&d_ctl.p->k, &k_, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    view_dirty = true;
}

// ------------------------------------------------------------------ column-block sharding
// Rendezvous: a POSIX shared-memory object created (zeroed) by the launcher, mapped by every rank:
//   [0, 768 * world)                 host-transport mailbox (MLP_TRANSPORT=host), registered with HIP
//   [768 * world, 896 * world)       one 128-byte rendezvous record per rank: HIP IPC handle of its device box
// Peer transport (default): every rank allocates its box in its OWN HBM (uncached, falling back to fine-grained),
// publishes the IPC handle, opens the peers' handles (peer access is enabled lazily by hipIpcOpenMemHandle), and
// the pivot kernels then write their 64-byte records straight into the peers' boxes over xGMI and poll locally.
namespace {
struct Rendezvous {  // 128 bytes
    uint64_t ready;  // 1: handle valid, 2: every peer handle opened by this rank
    int32_t device, pid;
    hipIpcMemHandle_t handle;  // 64 bytes
    // go-live check of the deferred sharding (Engine::golive_check): this rank's fingerprint of the replicated state, published under
    // fp_seq (the number of the check), acknowledged under fp_ack once every peer's fingerprint has been compared
    uint64_t fp_seq, fp_ack;
    uint64_t fp[4];  // pivots taken, nucleus size, hash of (basic_vars, nb_vars), bits of the objective
};
static_assert(sizeof(Rendezvous) == 128, "rendezvous record layout");
constexpr size_t kHostBoxBytesPerRank = sizeof(MailRec) * 2 * MAIL_KINDS;  // 768
bool wait_flag(volatile uint64_t* f, uint64_t want, double seconds) {
    const double t0 = now_s();
    while (__atomic_load_n(f, __ATOMIC_ACQUIRE) < want) {
        if (now_s() - t0 > seconds) return false;
        usleep(200);
    }
    return true;
}
// RCCL is loaded at run time (dlopen) and only when the rccl transport is asked for: the library has no link-time dependency
// on it, and a process that already holds a copy (torch ships one) shares that copy.
struct NcclId { char internal[128]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi& rccl_api() {
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (api.lib) return api;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;  // a copy the process already holds
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) throw MlpError(-3, std::string("rccl transport: cannot load librccl.so: ") + (dlerror() ? dlerror() : "not found"));
    api.GetUniqueId = reinterpret_cast<int (*)(NcclId*)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<int (*)(void**, int, NcclId, int)>(dlsym(h, "ncclCommInitRank"));
    api.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(h, "ncclAllGather"));
    api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy)
        throw MlpError(-3, "rccl transport: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    api.lib = h;
    return api;
}
void rccl_check(int rc, const char* what) {
    if (rc == 0) return;
    RcclApi& a = rccl_api();
    throw MlpError(-3, std::string("rccl transport: ") + what + " failed: " + (a.GetErrorString ? a.GetErrorString(rc) : std::to_string(rc).c_str()));
}
void rccl_destroy(void* comm) {
    if (comm) (void)rccl_api().CommDestroy(comm);
}
}  // namespace
void Engine::rccl_unique_id(void* out128) {
    NcclId id;
    std::memset(&id, 0, sizeof(id));
    rccl_check(rccl_api().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out128, &id, sizeof(id));
}
// ---- pump transport: see kernels.hip (k_mail_stage / k_mail_deliver) for the protocol on the device side
void Engine::enable_pump(int rank, int world, void* shm, size_t host_bytes, int backend, const void* rccl_id) {
    const size_t blk = (size_t)MAIL_PUMP_RECS * sizeof(MailRec);
    hipError_t e = hipExtMallocWithFlags(&own_box, host_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        own_box = nullptr;
        e = hipExtMallocWithFlags(&own_box, host_bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) throw MlpError(-3, std::string("enable_sharding: cannot allocate the device mailbox: ") + hipGetErrorString(e));
    HIPCHECK(hipMemset(own_box, 0, host_bytes));
    HIPCHECK(hipMalloc(&pump_stage_, blk * (size_t)world));
    HIPCHECK(hipMemset(pump_stage_, 0, blk * (size_t)world));
    HIPCHECK(hipHostMalloc(reinterpret_cast<void**>(&pump_host_), sizeof(unsigned long long) * (size_t)(world + 1), hipHostMallocDefault));
    std::memset(pump_host_, 0, sizeof(unsigned long long) * (size_t)(world + 1));
    HIPCHECK(hipStreamCreateWithFlags(&st_pump, hipStreamNonBlocking));
    HIPCHECK(hipDeviceSynchronize());
    pump_backend_ = backend;  // (from here on release_mailboxes() tears the pump down)
    pump_seq_ = 0;
    pump_gen_ = 0;
    for (int r = 0; r < MAX_WORLD; ++r) {
        peer_box[r] = nullptr;
        pump_stage_peer_[r] = nullptr;
    }
    d_mail = reinterpret_cast<MailRec*>(own_box);
    mail_fanout = 1;
    xb_cap_ = 0;  // (no row-sharded streaming pass: its vectors travel through peer stores only)
    if (backend == 1) {
        if (!rccl_id) throw MlpError(-1, "enable_sharding: the rccl transport needs rank 0's ncclUniqueId (mlp_rccl_unique_id)");
        NcclId id;
        std::memcpy(&id, rccl_id, sizeof(id));
        (void)rccl_api();  // dlopen before the readiness vote: a rank without librccl fails HERE, not inside the collective
        // ncclCommInitRank is a blocking collective without a timeout: a rank that failed above (allocation, dlopen) would leave the
        // others inside it forever.  Every rank therefore first publishes "ready" through the rendezvous object and waits — bounded —
        // for every peer's flag; a failed rank publishes all-ones (enable_sharding's catch) and every rank leaves through the same door.
        Rendezvous* rvr = reinterpret_cast<Rendezvous*>(static_cast<uint8_t*>(shm) + host_bytes);
        __atomic_store_n(&rvr[rank].ready, (uint64_t)1, __ATOMIC_RELEASE);
        for (int r = 0; r < world; ++r) {
            if (r == rank) continue;
            if (!wait_flag(&rvr[r].ready, 1, 120.0))
                throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " did not get ready for ncclCommInitRank within 120 s");
            if (__atomic_load_n(&rvr[r].ready, __ATOMIC_ACQUIRE) == ~(uint64_t)0)
                throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " failed before ncclCommInitRank");
        }
        rccl_check(rccl_api().CommInitRank(&rccl_comm_, world, id, rank), "ncclCommInitRank");
        ranks_share_device = false;  // (RCCL refuses two ranks on one device)
        transport = "RCCL: the ranks' mailbox records delivered by ncclAllGather over xGMI, pumped on a second stream while a batch of pivots is in flight";
    } else {
        // peer copies in place of the collective (ranks sharing one GPU, which RCCL refuses: how the pump protocol is tested)
        Rendezvous* rv = reinterpret_cast<Rendezvous*>(static_cast<uint8_t*>(shm) + host_bytes);
        Rendezvous* mine = rv + rank;
        HIPCHECK(hipIpcGetMemHandle(&mine->handle, pump_stage_));
        mine->pid = (int32_t)getpid();
        __atomic_store_n(&mine->ready, (uint64_t)1, __ATOMIC_RELEASE);
        for (int r = 0; r < world; ++r) {
            if (r == rank) {
                pump_stage_peer_[r] = pump_stage_;
                continue;
            }
            if (!wait_flag(&rv[r].ready, 1, 120.0))
                throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " did not publish its staging buffer within 120 s");
            if (__atomic_load_n(&rv[r].ready, __ATOMIC_ACQUIRE) == ~(uint64_t)0)
                throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " failed to set up its staging buffer");
            void* q = nullptr;
            hipError_t eo = hipIpcOpenMemHandle(&q, rv[r].handle, hipIpcMemLazyEnablePeerAccess);
            if (eo != hipSuccess)
                throw MlpError(-3, "enable_sharding: hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s staging buffer failed: " + hipGetErrorString(eo));
            pump_stage_peer_[r] = q;
        }
        __atomic_store_n(&mine->ready, (uint64_t)2, __ATOMIC_RELEASE);
        for (int r = 0; r < world; ++r)
            if (r != rank && !wait_flag(&rv[r].ready, 2, 120.0))
                throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " did not finish mapping the staging buffers within 120 s");
        ranks_share_device = true;
        transport = "pump: the ranks' mailbox records delivered by peer copies between staging buffers (HIP IPC) behind a shared-memory "
                    "barrier — the RCCL transport's protocol with the collective replaced, for ranks that share one GPU";
    }
}
void Engine::shm_barrier() {  // all ranks of the solve, through the counter at the head of the rendezvous object
    if (shard_world <= 1 || !mail_host) return;
    volatile uint64_t* cnt = reinterpret_cast<volatile uint64_t*>(mail_host);
    const uint64_t target = (++pump_gen_) * (uint64_t)shard_world;
    __atomic_add_fetch(cnt, (uint64_t)1, __ATOMIC_SEQ_CST);
    const double t0 = now_s();
    long spins = 0;
    while (__atomic_load_n(cnt, __ATOMIC_ACQUIRE) < target) {
        if ((++spins & 0xfff) == 0 && now_s() - t0 > 120.0) throw MlpError(-3, "pump transport: a peer rank did not reach the barrier within 120 s");
        __builtin_ia32_pause();
    }
}
// Deferred sharding goes live on the assumption that every rank is the same replica: same pivots taken, same basis, same nucleus.
// Nothing exchanged so far could have shown a divergence (replicas run without exchanges), and a sharded phase entered on
// inconsistent state would run on silently — every rank keeping only its own block of d / gamma from then on.  So the switch is
// guarded: every rank publishes a fingerprint of its replicated host-side state through the rendezvous object, waits (bounded) for
// its peers', and the solve fails loudly on EVERY rank if any two differ (or if a peer never arrives: it went live at another pivot).
void Engine::golive_check() {
    if (shard_world <= 1 || !mail_host) return;
    Rendezvous* rv = reinterpret_cast<Rendezvous*>(static_cast<uint8_t*>(mail_host) + kHostBoxBytesPerRank * (size_t)shard_world);
    uint64_t h = 1469598103934665603ull;  // FNV-1a over the two index sets
    auto mix = [&h](const std::vector<int>& a) {
        for (int x : a) {
            h ^= (uint64_t)(uint32_t)x;
            h *= 1099511628211ull;
        }
    };
    mix(h_basic_vars);
    mix(h_nb_vars);
    const double obj = cur_obj_val();
    uint64_t ob = 0;
    std::memcpy(&ob, &obj, sizeof(ob));
    uint64_t fp[4] = {(uint64_t)lifetime_pivots, (uint64_t)(int64_t)k_, h, ob};
    {   // test hook: MLP_TEST_GOLIVE_SKEW=<rank> makes that rank publish a perturbed fingerprint (tests/test_dist_gpu.py)
        const char* sk = std::getenv("MLP_TEST_GOLIVE_SKEW");
        if (sk && std::atoi(sk) == shard_rank) fp[2] ^= 1ull;
    }
    Rendezvous* mine = rv + shard_rank;
    const uint64_t seq = ++golive_seq_;
    for (int j = 0; j < 4; ++j) mine->fp[j] = fp[j];
    __atomic_store_n(&mine->fp_seq, seq, __ATOMIC_RELEASE);
    std::string bad;
    for (int r = 0; r < shard_world; ++r) {
        if (r == shard_rank) continue;
        if (!wait_flag(&rv[r].fp_seq, seq, 120.0)) {
            bad = "rank " + std::to_string(r) + " did not reach the go-live point of the deferred sharding within 120 s (it diverged or stopped)";
            break;
        }
        for (int j = 0; j < 4 && bad.empty(); ++j)
            if (rv[r].fp[j] != fp[j]) {
                static const char* what[4] = {"pivots taken", "nucleus size", "basis (hash of basic_vars / nb_vars)", "objective bits"};
                bad = std::string("ranks ") + std::to_string(shard_rank) + " and " + std::to_string(r) + " are not the same replica at the go-live point of the "
                      "deferred sharding: " + what[j] + " differ (" + std::to_string(fp[j]) + " vs " + std::to_string(rv[r].fp[j]) + ")";
            }
        if (!bad.empty()) break;
    }
    // acknowledge whatever the outcome, so that no peer overwrites a fingerprint another rank is still reading (a later check of the
    // same rendezvous object) and a failing rank does not leave the others waiting for its acknowledgement
    __atomic_store_n(&mine->fp_ack, seq, __ATOMIC_RELEASE);
    if (!bad.empty()) throw MlpError(-3, "enable_sharding / go-live: " + bad);
    for (int r = 0; r < shard_world; ++r)
        if (r != shard_rank && !wait_flag(&rv[r].fp_ack, seq, 120.0))
            throw MlpError(-3, "enable_sharding / go-live: rank " + std::to_string(r) + " did not acknowledge the fingerprint check within 120 s");
    golive_checks_ += 1;
}
void Engine::pump_round(bool done_local, bool* all_done) {
    const int world = shard_world, rank = shard_rank;
    const size_t blk = (size_t)MAIL_PUMP_RECS * sizeof(MailRec);
    uint8_t* stage = static_cast<uint8_t*>(pump_stage_);
    pump_host_[world] = done_local ? pump_seq_ : pump_seq_ - 1;  // this rank's "my batch is done" word travels with its records
    HIPCHECK(hipMemcpyAsync(stage + (size_t)rank * blk + (size_t)(MAIL_PUMP_RECS - 1) * sizeof(MailRec), &pump_host_[world], sizeof(unsigned long long),
                            hipMemcpyHostToDevice, st_pump));
    launch_mail_stage(hview, stage, st_pump);
    if (pump_backend_ == 1) {
        rccl_check(rccl_api().AllGather(stage + (size_t)rank * blk, stage, blk, /* ncclInt8 */ 0, rccl_comm_, st_pump), "ncclAllGather");
    } else {
        HIPCHECK(hipStreamSynchronize(st_pump));
        shm_barrier();  // every rank's block is staged
        for (int r = 0; r < world; ++r)
            if (r != rank)
                HIPCHECK(hipMemcpyAsync(stage + (size_t)r * blk, static_cast<uint8_t*>(pump_stage_peer_[r]) + (size_t)r * blk, blk, hipMemcpyDeviceToDevice, st_pump));
        HIPCHECK(hipStreamSynchronize(st_pump));
        shm_barrier();  // nobody stages the next round while a peer still copies this one
    }
    launch_mail_deliver(hview, stage, st_pump);
    HIPCHECK(hipMemcpy2DAsync(pump_host_, sizeof(unsigned long long), stage + (size_t)(MAIL_PUMP_RECS - 1) * sizeof(MailRec), blk, sizeof(unsigned long long),
                              (size_t)world, hipMemcpyDeviceToHost, st_pump));
    HIPCHECK(hipStreamSynchronize(st_pump));
    bool all = true;
    for (int r = 0; r < world; ++r) all = all && pump_host_[r] >= pump_seq_;
    *all_done = all;
    pump_rounds_ += 1;
}
// The exchanges of the kernels just enqueued on `st` complete only if the pump runs: keep delivering until EVERY rank's stream has
// drained (each round is a collective, and the decision to stop is taken from the gathered words: all ranks stop together).
void Engine::pump_until_idle() {
    if (!pump_backend_) return;
    pump_seq_ += 1;
    const double t0 = now_s();
    double next_report = 1.0;
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q != hipSuccess && q != hipErrorNotReady) HIPCHECK(q);
        bool all = false;
        pump_round(q == hipSuccess, &all);
        if (all) break;
        static const bool dbg = std::getenv("MLP_PUMP_DEBUG") != nullptr;
        if (dbg && now_s() - t0 > next_report) {
            next_report += 1.0;
            std::fprintf(stderr, "[mlp pump] rank %d seq %llu: %.1f s, %llu rounds so far, local stream %s, words", shard_rank, pump_seq_, now_s() - t0,
                         (unsigned long long)pump_rounds_, q == hipSuccess ? "idle" : "busy");
            for (int r = 0; r < shard_world; ++r) std::fprintf(stderr, " %llu", pump_host_[r]);
            std::fprintf(stderr, "\n");
        }
        if (now_s() - t0 > 600.0) throw MlpError(-3, "pump transport: the batch did not drain on every rank within 600 s");
    }
}
void Engine::release_mailboxes() {
    for (int r = 0; r < MAX_WORLD; ++r) {
        if (peer_box[r] && r != shard_rank) (void)hipIpcCloseMemHandle(peer_box[r]);
        peer_box[r] = nullptr;
    }
    if (own_box) (void)hipFree(own_box);
    own_box = nullptr;
    if (pump_backend_) {
        if (st_pump) (void)hipStreamSynchronize(st_pump);
        for (int r = 0; r < MAX_WORLD; ++r) {
            if (pump_stage_peer_[r] && pump_stage_peer_[r] != pump_stage_) (void)hipIpcCloseMemHandle(pump_stage_peer_[r]);
            pump_stage_peer_[r] = nullptr;
        }
        if (rccl_comm_) rccl_destroy(rccl_comm_);
        rccl_comm_ = nullptr;
        if (pump_stage_) (void)hipFree(pump_stage_);
        pump_stage_ = nullptr;
        if (pump_host_) (void)hipHostFree(pump_host_);
        pump_host_ = nullptr;
        if (st_pump) (void)hipStreamDestroy(st_pump);
        st_pump = nullptr;
        pump_backend_ = 0;
    }
    if (mail_host) {
        if (mail_registered) (void)hipHostUnregister(mail_host);
        (void)munmap(mail_host, mail_bytes);
    }
    mail_host = nullptr;
    mail_registered = false;
    d_mail = nullptr;
}
void Engine::enable_sharding(int rank, int world, const char* shm_name, const char* transport_name, const void* rccl_id) {
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world)
        throw MlpError(-1, "enable_sharding: bad rank/world (at most " + std::to_string(MAX_WORLD) + " ranks)");
    HIPCHECK(hipStreamSynchronize(st));
    // The compact factor under sharding: the solves run replicated on every rank (fixed-order sums), the column-block kernels do the rest —
    // pivot for pivot identical to the unsharded run (round 5).  It is PATHOLOGICAL only when ranks share a device: k_fac_solve is a
    // persistent kernel with grid barriers, one workgroup per CU, and two of them from two processes time-slice the device while each
    // one's peers wait in a mailbox spin (3.9 s per pivot with two ranks on one GPU).  The gate therefore follows `ranks_share_device`,
    // which is known only once the transport is up: see the end of this function (MLP_FACTOR_SHARED_DEVICE=1 keeps the factor even then:
    // the oversubscribed correctness test).
    const bool had_factor = fac_on_;
    release_mailboxes();
    std::string tname = transport_name ? transport_name : "";
    if (tname.empty() && std::getenv("MLP_TRANSPORT")) tname = std::getenv("MLP_TRANSPORT");
    const int pump = tname == "rccl" ? 1 : (tname == "pump" ? 2 : 0);
    if (!tname.empty() && !pump && tname != "peer" && tname != "host")
        throw MlpError(-1, "enable_sharding: unknown transport '" + tname + "' (peer, host, rccl, pump)");
    if (world == 1 && !pump) {  // (the pump transports accept a world of one: the collective path is then exercised end to end with a single rank)
        shard_rank = 0; shard_world = 1;
        view_dirty = true;
        return;
    }
    const bool host_transport = !pump && tname == "host";
    const size_t host_bytes = kHostBoxBytesPerRank * (size_t)world;
    const size_t bytes = host_bytes + sizeof(Rendezvous) * (size_t)world;
    int fd = shm_open(shm_name, O_RDWR, 0600);
    if (fd < 0) throw MlpError(-1, std::string("enable_sharding: shm_open failed for ") + shm_name);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) {
        close(fd);
        throw MlpError(-1, "enable_sharding: mailbox object too small (need 896 * world bytes)");
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) throw MlpError(-1, "enable_sharding: mmap failed");
    mail_host = p; mail_bytes = bytes;
    shard_rank = rank;  // release_mailboxes() skips this rank's own entry of peer_box
    try {
        if (pump) {
            enable_pump(rank, world, p, host_bytes, pump, rccl_id);
        } else if (host_transport) {
            HIPCHECK(hipHostRegister(p, bytes, hipHostRegisterMapped));
            mail_registered = true;
            void* dp = nullptr;
            HIPCHECK(hipHostGetDevicePointer(&dp, p, 0));
            d_mail = reinterpret_cast<MailRec*>(dp);
            for (int r = 0; r < world; ++r) peer_box[r] = nullptr;
            mail_fanout = 1;
            ranks_share_device = true;  // unknown with this transport: take the form that needs no co-residency
            transport = "host-mapped shared-memory mailbox (PCIe)";
        } else {
            // one allocation: the mailbox ([kind][parity][rank] records) followed by the exchange buffer of the
            // row-sharded streaming pass ([parity][rank][tau_K | v_K partial][xb_cap] doubles), so that the one IPC
            // mapping a peer opens covers both
            xb_cap_ = (((size_t)std::max(m_, 1) + 1023) / 1024) * 1024 + 64;
            const size_t box_bytes = host_bytes + sizeof(double) * 2 * (size_t)world * 2 * xb_cap_;
            hipError_t e = hipExtMallocWithFlags(&own_box, box_bytes, hipDeviceMallocUncached);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                own_box = nullptr;
                e = hipExtMallocWithFlags(&own_box, box_bytes, hipDeviceMallocFinegrained);
            }
            if (e != hipSuccess) throw MlpError(-3, std::string("enable_sharding: cannot allocate the device mailbox: ") + hipGetErrorString(e));
            HIPCHECK(hipMemset(own_box, 0, box_bytes));
            HIPCHECK(hipDeviceSynchronize());
            Rendezvous* rv = reinterpret_cast<Rendezvous*>(static_cast<uint8_t*>(p) + host_bytes);
            Rendezvous* mine = rv + rank;
            HIPCHECK(hipIpcGetMemHandle(&mine->handle, own_box));
            int dev = 0;
            HIPCHECK(hipGetDevice(&dev));
            {   // physical identity of the device (ordinals differ between processes with different visibility masks)
                int dom = 0, bus = 0, did = 0;
                (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, dev);
                (void)hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev);
                (void)hipDeviceGetAttribute(&did, hipDeviceAttributePciDeviceId, dev);
                mine->device = ((dom & 0xffff) << 16) | ((bus & 0xff) << 8) | (did & 0xff);
            }
            mine->pid = (int32_t)getpid();
            __atomic_store_n(&mine->ready, (uint64_t)1, __ATOMIC_RELEASE);
            for (int r = 0; r < world; ++r) {
                if (r == rank) {
                    peer_box[r] = own_box;
                    continue;
                }
                if (!wait_flag(&rv[r].ready, 1, 120.0))
                    throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " did not publish its mailbox handle within 120 s");
                if (__atomic_load_n(&rv[r].ready, __ATOMIC_ACQUIRE) == ~(uint64_t)0)
                    throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " failed to set up its mailbox");
                if (rv[r].pid == mine->pid) throw MlpError(-1, "enable_sharding: two ranks in one process are not supported");
                void* q = nullptr;
                hipError_t eo = hipIpcOpenMemHandle(&q, rv[r].handle, hipIpcMemLazyEnablePeerAccess);
                if (eo != hipSuccess)
                    throw MlpError(-3, "enable_sharding: hipIpcOpenMemHandle of rank " + std::to_string(r) + "'s mailbox (device " +
                                           std::to_string(rv[r].device) + " from device " + std::to_string(dev) + ") failed: " +
                                           hipGetErrorString(eo));
                peer_box[r] = q;
            }
            __atomic_store_n(&mine->ready, (uint64_t)2, __ATOMIC_RELEASE);
            // nobody may start posting before every rank has mapped every box it will be written from / polled in
            for (int r = 0; r < world; ++r)
                if (r != rank && !wait_flag(&rv[r].ready, 2, 120.0))
                    throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " did not finish mapping the mailboxes within 120 s");
            for (int r = 0; r < world; ++r)
                if (r != rank && __atomic_load_n(&rv[r].ready, __ATOMIC_ACQUIRE) == ~(uint64_t)0)
                    throw MlpError(-3, "enable_sharding: rank " + std::to_string(r) + " failed while mapping the mailboxes");
            d_mail = reinterpret_cast<MailRec*>(own_box);
            mail_fanout = world;
            bool same_dev = true;
            ranks_share_device = false;
            for (int r = 0; r < world; ++r) {
                same_dev = same_dev && rv[r].device == mine->device;
                if (r != rank && rv[r].device == mine->device) ranks_share_device = true;
            }
            transport = same_dev ? "device-resident mailboxes mapped through HIP IPC (all ranks on one GPU)"
                                 : "device-resident mailboxes in each GPU's HBM, written by the peers over xGMI (HIP IPC peer mappings)";
        }
    } catch (...) {
        // tell the peers at once (they would otherwise wait up to 120 s per missing flag): ready = all ones means "failed"
        if (!host_transport && mail_host) {
            Rendezvous* rvf = reinterpret_cast<Rendezvous*>(static_cast<uint8_t*>(mail_host) + host_bytes);
            __atomic_store_n(&rvf[rank].ready, ~(uint64_t)0, __ATOMIC_RELEASE);
        }
        release_mailboxes();  // (nobody posts into this rank's box before every rank has reached ready = 2)
        shard_rank = 0; shard_world = 1;
        view_dirty = true;
        throw;
    }
    shard_rank = rank; shard_world = world;
    shard_live_ = true;  // (the handshake below runs through the live view; deferral is decided after it)
    view_dirty = true;
    // Handshake: before any pivot depends on it, every rank posts one record to every box and waits for all of them
    // with a short bound (~2 s).  A transport that maps but does not deliver (no peer-to-peer route, non-coherent
    // mapping) is detected HERE — the caller can then fall back to the host mailbox — not as a 10-second stall
    // inside the first sharded pivot.
    try {
        sync_view();
        DevBuf<int> flag;
        flag.ensure(2, 0, st);
        HIPCHECK(hipMemsetAsync(flag.p, 0, 2 * sizeof(int), st));
        launch_mail_handshake(hview, flag.p, st);
        if (pump_backend_) pump_until_idle();  // (the handshake records travel through the pump like every other record)
        int hf[2] = {0, 0};
        HIPCHECK(hipMemcpyAsync(hf, flag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (hf[0] != 1)
            throw MlpError(-3, "enable_sharding: the mailbox handshake did not complete (rank " + std::to_string(rank) + " heard from " +
                                   std::to_string(hf[1]) + " of " + std::to_string(world) + " ranks): transport '" + transport + "' does not deliver");
    } catch (...) {
        // The peers may still be posting their handshake records into this rank's box through their IPC mappings: it is
        // NOT freed here (release_mailboxes() runs at the next enable_sharding call or when the Solution goes away);
        // the view stops pointing at it at once.
        shard_rank = 0; shard_world = 1;
        shard_live_ = false;
        view_dirty = true;
        throw;
    }
    if (had_factor && fac_on_ && world > 1 && !fac_allowed_under_sharding()) {
        fac_leave();  // (ranks sharing a device, or a transport that cannot tell: back to the explicit inverse)
        view_dirty = true;
    }
    {   // deferred sharding (engine.h): replicas until the tableau row becomes a pass over A; MLP_SHARD_DEFER=0 shards from the first pivot
        const char* sd = std::getenv("MLP_SHARD_DEFER");
        shard_defer_ = !(sd && sd[0] == '0');
        shard_live_ = !(shard_defer_ && world > 1 && !pump_backend_);
        view_dirty = true;
    }
}

// ------------------------------------------------------------------ Solver::try_new (solver.rs:108-369)
void Engine::try_new(const ProblemData& pd) {
    direction = pd.direction;
    num_vars = (int)pd.obj.size();
    const int n = num_vars;
    h_lo = pd.lo;
    h_hi = pd.hi;
    std::vector<double> xN(n), d(n), gamma;
    std::vector<uint8_t> flags(n);
    double obj_val = 0.0;
    dual_feasible = true;
    h_nb_vars.resize(n);
    h_var_loc.resize(n);
    for (int v = 0; v < n; ++v) {  // solver.rs:133-187: initial non-basic values, aiming at dual feasibility
        double mn = h_lo[v], mx = h_hi[v], c = pd.obj[v];
        if (mn > mx) throw LpFail{1};
        h_nb_vars[v] = v;
        h_var_loc[v] = -1 - v;
        double init;
        if (mn == mx) init = mn;
        else if (std::isinf(mn) && std::isinf(mx)) {
            if (c != 0.0) dual_feasible = false;
            init = 0.0;
        } else if (c > 0.0) {
            if (std::isfinite(mn)) init = mn;
            else { dual_feasible = false; init = mx; }
        } else if (c < 0.0) {
            if (std::isfinite(mx)) init = mx;
            else { dual_feasible = false; init = mn; }
        } else if (std::isfinite(mn)) init = mn;
        else init = mx;
        xN[v] = init;
        obj_val += init * c;
        flags[v] = (uint8_t)((init == mn ? NB_AT_MIN : 0) | (init == mx ? NB_AT_MAX : 0));
    }

    // rows: slack per non-empty constraint, all slacks basic (solver.rs:198-253)
    h_rptr.assign(1, 0);
    h_rcol.clear();
    h_rval.clear();
    h_rhs.clear();
    std::vector<double> xB, loB, hiB;
    std::vector<const Constraint*> kept;
    for (const Constraint& c : pd.cons) {
        if (c.idx.empty()) {
            bool taut = c.op == 0 ? (0.0 == c.rhs) : c.op == 1 ? (0.0 <= c.rhs) : (0.0 >= c.rhs);
            if (taut) continue;
            throw LpFail{1};
        }
        kept.push_back(&c);
    }
    m_ = (int)kept.size();
    N_ = n + m_;
    h_obj = pd.obj;
    h_obj.resize(N_, 0.0);
    h_lo.resize(N_);
    h_hi.resize(N_);
    h_basic_vars.resize(m_);
    h_var_loc.resize(N_);
    for (int r = 0; r < m_; ++r) {
        const Constraint& c = *kept[r];
        double smin = c.op == 1 ? 0.0 : c.op == 2 ? -INF : 0.0;
        double smax = c.op == 1 ? INF : 0.0;
        int sv = n + r;
        h_lo[sv] = smin;
        h_hi[sv] = smax;
        loB.push_back(smin);
        hiB.push_back(smax);
        h_basic_vars[r] = sv;
        h_var_loc[sv] = r;
        double lhs = 0.0;
        for (size_t p = 0; p < c.idx.size(); ++p) {
            lhs += c.val[p] * xN[c.idx[p]];
            h_rcol.push_back(c.idx[p]);
            h_rval.push_back(c.val[p]);
        }
        h_rcol.push_back(sv);  // slack coefficient is always +1 (solver.rs:250)
        h_rval.push_back(1.0);
        h_rptr.push_back((int)h_rcol.size());
        h_rhs.push_back(c.rhs);
        xB.push_back(c.rhs - lhs);
    }
    build_csc();

    primal_feasible = true;  // solver.rs:255-259
    for (int r = 0; r < m_; ++r)
        if (!(xB[r] >= loB[r] && xB[r] <= hiB[r])) { primal_feasible = false; break; }
    bool need_artificial = !primal_feasible && !dual_feasible;  // solver.rs:261
    enable_dse = true;                                          // solver.rs:263
    enable_pse = !dual_feasible;                                // solver.rs:272
    for (int c = 0; c < n; ++c) {                               // solver.rs:281-300
        uint8_t f = flags[c];
        bool amin = f & NB_AT_MIN, amax = f & NB_AT_MAX;
        if (need_artificial) d[c] = (amin && !amax) ? 1.0 : (amax && !amin) ? -1.0 : 0.0;
        else d[c] = h_obj[c];
    }
    gamma.assign(n, 1.0);
    if (enable_pse)
        for (int c = 0; c < n; ++c) {
            double s = 0.0;
            for (int p = h_cptr[c]; p < h_cptr[c + 1]; ++p) s += h_cval[p] * h_cval[p];
            gamma[c] = s + 1.0;
        }
    double cur_obj = need_artificial ? 0.0 : obj_val;  // solver.rs:302

    // basis inverse: every initial basic column is a slack singleton => empty nucleus (the
    // reference factorises the identity here, solver.rs:304-317)
    k_ = 0;
    h_kslot_of_pos.assign(m_, -1);
    h_srow_of_pos.resize(m_);
    h_sdiag_of_pos.assign(m_, 1.0);
    h_kslot_of_row.assign(m_, -1);
    h_pos_of_srow.resize(m_);
    for (int r = 0; r < m_; ++r) h_srow_of_pos[r] = h_pos_of_srow[r] = r;
    h_nb_fixed.assign(n, 0);
    nnz_nonbasic = 0;
    for (int c = 0; c < n; ++c) nnz_nonbasic += col_nnz(c);

    // ---- upload
    upload_matrix();
    int m_keep = m_;
    m_ = 0;  // nothing to preserve
    alloc_row_buffers(m_keep);
    m_ = m_keep;
    d_var_loc.upload(h_var_loc, st);
    d_basic_vars.upload(h_basic_vars, st);
    d_xB.upload(xB, st); d_loB.upload(loB, st); d_hiB.upload(hiB, st);
    std::vector<double> beta(m_, 1.0);
    d_beta.upload(beta, st);
    d_nb_vars.upload(h_nb_vars, st);
    d_d.upload(d, st); d_xN.upload(xN, st); d_gamma.upload(gamma, st); d_nbflags.upload(flags, st);
    d_alpha_r.ensure(n, 0, st); d_helper.ensure(n, 0, st); d_nb_rng.ensure(n, 0, st);
    ensure_red();
    d_ticket.ensure(TK_WORDS, 0, st);
    HIPCHECK(hipMemsetAsync(d_ticket.p, 0, d_ticket.cap * sizeof(unsigned), st));
    d_ctl.ensure(1, 0, st);
    std::memset(h_ctl, 0, sizeof(Ctl));
    h_ctl->it.obj = cur_obj;
    h_ctl->it.status = ITER_NONE;
    h_ctl->up.kase = -1;
    h_ctl->ratio_spin_limit = ratio_spin_limit;
    h_ctl->kprof_on = std::getenv("MLP_KPROF") ? std::atoi(std::getenv("MLP_KPROF")) : 0;  // 1: kernel timeline marks; 2..5: per-rank quantity in the records (diagnostics)
    HIPCHECK(hipMemcpyAsync(d_ctl.p, h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, st));
    ensure_nucleus_cap(256);
    push_maps();
    sync_view();
    launch_init_nb_rng(hview, geom(), st);
    HIPCHECK(hipStreamSynchronize(st));
    values_dirty = true;
    if (fac_mode == 1 && m_ > 0) (void)fac_enter();  // MLP_FACTOR=1: the compact factor from the slack basis on (one level)
}

// ------------------------------------------------------------------ values / objective
void Engine::fetch_values() {
    if (!values_dirty) return;
    h_xB.resize(m_);
    h_xN.resize(num_vars);
    if (m_) HIPCHECK(hipMemcpyAsync(h_xB.data(), d_xB.p, sizeof(double) * m_, hipMemcpyDeviceToHost, st));
    if (num_vars) HIPCHECK(hipMemcpyAsync(h_xN.data(), d_xN.p, sizeof(double) * num_vars, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    values_dirty = false;
}
double Engine::get_value(int var) {  // solver.rs:371-376
    fetch_values();
    int loc = h_var_loc[var];
    return loc >= 0 ? h_xB[loc] : h_xN[-1 - loc];
}
void Engine::get_values(double* out, int n) {
    fetch_values();
    for (int v = 0; v < n; ++v) {
        int loc = h_var_loc[v];
        out[v] = loc >= 0 ? h_xB[loc] : h_xN[-1 - loc];
    }
}
double Engine::cur_obj_val() {
    pull_ctl();
    return h_ctl->it.obj;
}

// ------------------------------------------------------------------ one iteration = a fixed kernel sequence
// primal: choose_pivot + pivot (solver.rs:695-853, 1023-1104); dual: solver.rs:529-533.
// Stage order of one iteration (the engine-level C ABI steps through the same table):
//   primal: FTRAN -> RATIO (+ BTRAN head, plan) -> BTRAN -> BASIS (fused W pass + tails) -> ROW -> APPLY
//   dual  : BTRAN -> ROW -> RATIO (+ FTRAN head) -> FTRAN (+ plan) -> BASIS -> APPLY
static const int kStageOrder[2][6] = {{STAGE_FTRAN, STAGE_RATIO, STAGE_BTRAN, STAGE_BASIS, STAGE_ROW, STAGE_APPLY},
                                      {STAGE_BTRAN, STAGE_ROW, STAGE_RATIO, STAGE_FTRAN, STAGE_BASIS, STAGE_APPLY}};
void Engine::launch_stage(int phase, int stage, bool with_events) {
    const DevView& dv = hview;
    const Geom g = geom();
    const int pse = enable_pse ? 1 : 0;
    const bool lazy = lazy_now(phase);                       // primal iteration: no tau, no beta update (ensure_beta() later)
    const int dse = (enable_dse && !lazy) ? 1 : 0;
    const int wtau = lazy ? 0 : 1;
    // banded sweep, primal iteration: k_update_pivot sums the per-band partials itself (no combine launch);
    // the host-paced stepping API keeps the separate combine so that row_coeffs is readable after STAGE_ROW
    const int inl = (dv.banded && phase == 0 && !stepping && !g.str) ? 1 : 0;
    if (dv.fac_on) {  // compact factor of the basis (factor.inc): the same stages over level-scheduled solves
        launch_stage_fac(phase, stage, with_events);
        return;
    }
    // small nucleus (first capacity of the inverse), lazy primal iteration: BTRAN, pass over W, v tail and touched-column list are
    // ONE launch (k_small_basis) issued at the BASIS stage; the BTRAN stage is empty.  MLP_SMALL_BASIS=0: the three launches.
    // ... and for a nucleus of a few dozen columns the whole chain FTRAN -> ratio test -> BTRAN -> inverse update -> touched columns -> partition
    // change is ONE launch of one workgroup issued at the FTRAN stage (k_primal_head); the RATIO, BTRAN and BASIS stages are empty
    const bool phead = phase == 0 && pse && lazy && !stepping && shard_world == 1 && primal_head_supported(dv, g);
    const bool smallb = !phead && phase == 0 && pse && lazy && !stepping && !shard_is_live() && small_basis_supported(dv, g);
    // large nucleus, lazy primal iteration: t_K = alpha_K - F^T y_S rides in the ratio test's launch (blocks behind the ratio blocks);
    // the FTRAN's push combine leaves y_S by row, the BTRAN launch forms rho_K only.  MLP_TK_RIDE=0: t_K in the BTRAN launch.
    const bool tkr = phase == 0 && pse && lazy && !stepping && shard_world == 1 && !smallb && !g.head_fused && !dv.pb_det && tk_rides_ratio(dv, g);
    const bool tkr_s = smallb && tk_rides_ratio_small(dv, g);  // small nucleus: t_K rides in the ratio launch too (y_S on the fly)
    // ... and rho_K rides behind the v tail of the pass (k_post_fused): the BTRAN stage is then empty.  MLP_RK_RIDE=0: its own launch.
    const bool rkr = tkr && rk_rides_post(dv, g);
    // ... and the F product of the FTRAN is PULLED inside the ratio test's launch (fpull.inc): no blocked push, no combine; the gather also
    // leaves alpha_K by variable and pushes the few columns that entered the basis since the packed copy was built.  MLP_FPULL=0: the push.
    // A sharded solve takes it too (every rank pulls the whole product, as it pushed it before: the FTRAN is replicated): the gather with the
    // head inside, the two launches of the pull; t_K rides in the second one, rho_K keeps its BTRAN launch (the v tail waits for the exchange).
    const bool fpl_base = g.fp && phase == 0 && pse && lazy && !stepping && !smallb && !g.head_fused && max_col_nnz_ <= HEAD_LIST_CAP && fpull_supported(dv, g);
    const bool fpl = fpl_base && ((tkr && ftran_head_rides_gather(dv, g)) || (shard_world > 1 && dv.lrJ > 0 && dv.pb_on && !dv.det_pull && dv.rowinfo != nullptr));
    if (stage == STAGE_BASIS) touch_done = false;
    // The pricing decision (q for primal, r for dual) is already in Ctl: it was taken by the
    // previous iteration's update kernel, or by the standalone pricing kernel at batch start.
    switch (stage) {
    case STAGE_FTRAN:
        if (with_events) HIPCHECK(hipEventRecord(ev[6], st));
        if (phead) {
            if (with_events) arm_kernel_timing(2, ev[2], ev[3]);  // (sampled iteration: the head is timed kernel-exactly in the slot of the pass over W, which it contains)
            launch_primal_head(dv, g, st);
            if (with_events) arm_kernel_timing(2, nullptr, nullptr);
            touch_done = true;
            if (with_events) HIPCHECK(hipEventRecord(ev[7], st));
            break;
        }
        if (phase == 0 && g.head_fused) {
            launch_ftran_fused(dv, g, 1, st);              // K2 head inside the gather kernel (one launch)
        } else if (phase == 0 && !stepping && max_col_nnz_ <= HEAD_LIST_CAP && (fpl || ftran_head_rides_gather(dv, g))) {
            launch_ftran_gather_lrh(dv, g, st, tkr ? 1 : 0, fpl ? 1 : 0);  // delayed-update mode: the head inside the gather too (round 5)
        } else {
            if (phase == 0) launch_ftran_prep(dv, 1, st);  // K2 head: entering column scalars, singleton rows, list
            launch_ftran_gather(dv, g, st, tkr ? 1 : 0);   // K2: alpha_q = B^-1 a_q (dual: the head ran in RATIO); t_K ride: + y_S by row
        }
        if (with_events && !fpl) HIPCHECK(hipEventRecord(ev[7], st));  // (fpl: the F product of this FTRAN runs inside the ratio test's launch — stamped there)
        if (phase == 1) {
            launch_post_ftran(dv, g, pse, st);         // alpha_sq, y_S, partition plan
            if (pse) launch_btran_rhs(dv, g, st);      // tK
        }
        break;
    case STAGE_RATIO:
        if (phead) break;
        if (fpl) {
            // pull of -F alpha_K (+ y_S) with K5 p1 | K5 p2 (+ K3 head + plan) | t_K; sampled iteration: the FTRAN bracket closes behind the
            // first of the two launches, which completes alpha_q
            launch_fpull_ratio(dv, g, st, with_events ? ev[7] : nullptr);
        } else if (phase == 0) launch_ratio_primal(dv, g, pse, st, tkr ? 1 : (tkr_s ? 2 : 0));  // K5 p1 (+ ||alpha||^2, y_S), p2 (+ K3 head + plan) [| t_K]
        else launch_ratio_dual(dv, g, st, ar_built_ ? 1 : 0);  // K7 p1, p2 (+ K2 head); over the listed non-zeros of alpha_r when the tableau row listed them
        break;
    case STAGE_BTRAN:
        if (phead || smallb || rkr) break;
        if (phase == 1 && g.head_fused) {
            launch_btran_fused(dv, g, 0, 1, st);                  // K3 head inside the BTRAN kernel (one launch)
        } else {
            if (phase == 1) launch_btran_prep(dv, 1, 0, st);      // K3 head (device-driven by it.r)
            launch_btran(dv, g, (phase == 0 && !tkr && !fpl) ? pse : 0, st);    // K3: rho, rK, ||rho||^2  |  tK = alpha_K - F^T y_S (unless it rode in the ratio launch)
        }
        break;
    case STAGE_BASIS:
        if (phead) {
            touch_done = true;  // (the head listed the touched columns)
            break;
        }
        if (smallb) {
            if (with_events) arm_kernel_timing(2, ev[2], ev[3]);  // (sampled iteration: the slot of the pass over the nucleus inverse)
            launch_small_basis(dv, g, st, tkr_s ? 0 : 1);
            if (with_events) arm_kernel_timing(2, nullptr, nullptr);
            touch_done = true;
            break;
        }
        if (with_events) {  // sampled iteration: the pass over the nucleus inverse and the fold are timed kernel-exactly
            arm_kernel_timing(2, ev[2], ev[3]);
            arm_kernel_timing(3, ev[10], ev[11]);
        }
        launch_fused_w(dv, g, pse, st, wtau);                 // tauK / vK partials + eta update of W
        if (with_events) {
            arm_kernel_timing(2, nullptr, nullptr);
            arm_kernel_timing(3, nullptr, nullptr);
        }
        // tau by position (F push)  |  v reduce + scatter  |  (sparse tableau row, primal: its touched-column list)
        touch_done = launch_post_fused(dv, g, pse, st, 0, 0, wtau, (g.str && phase == 0) ? 1 : 0, rkr ? 1 : 0) != 0;
        break;
    case STAGE_ROW:
        ar_built_ = false;
        if (g.str) {  // small nucleus: the columns that meet supp(rho) only (k_row_touch + k_row_pull)
            if (with_events) HIPCHECK(hipEventRecord(ev[0], st));  // (sampled iteration: the two launches bracketed together)
            if (phead && update_pulls_inside(dv, g)) {  // the update kernel's workgroups pull the touched columns of their own positions
                if (with_events) HIPCHECK(hipEventRecord(ev[1], st));
                break;
            }
            if (phase == 0) launch_row_sparse(dv, g, pse ? 1 : 0, phead ? 0 : 1, touch_done ? 0 : 1, st);  // (the head applied the partition change itself)
            else launch_row_sparse(dv, g, 0, 0, 1, st);
            if (with_events) HIPCHECK(hipEventRecord(ev[1], st));
            break;
        }
        if (with_events) arm_kernel_timing(1, ev[0], ev[1]);  // sampled iteration: the sweep kernel is timed kernel-exactly
        if (phase == 0) launch_sweep(dv, g, pse ? 1 : 0, 1, st, inl);  // K4 (+ PSE helper in the same pass)  |  partition change
        else ar_built_ = launch_sweep(dv, g, 0, 0, st, 0, 1);  // K4: alpha_r = rho^T N (+ the list of its non-zeros: CSC-pull form)
        if (with_events) arm_kernel_timing(1, nullptr, nullptr);
        break;
    case STAGE_APPLY:
        if (phase == 1 && pse) {  // dual path: PSE helper  |  partition change
            if (g.str) launch_row_sparse(dv, g, 2, 1, 0, st);  // (on the columns the ROW stage listed)
            else launch_sweep(dv, g, 2, 1, st);
        }
        if (with_events) HIPCHECK(hipEventRecord(ev[4], st));
        // K8 + zero the work vectors + price the next iteration (dual path without PSE: | partition change)
        launch_update_pivot(dv, g, phase, dse, pse, st, inl, (phase == 1 && !pse) ? 1 : 0, (phead && g.str && update_pulls_inside(dv, g)) ? (head_applies(dv, g) ? 2 : 1) : 0);
        if (with_events) HIPCHECK(hipEventRecord(ev[5], st));
        break;
    default:
        throw MlpError(-1, "unknown stage");
    }
}
static bool use_graph_env() {
    const char* ng = std::getenv("MLP_NO_GRAPH");
    return !(ng && ng[0] == '1');
}
void Engine::record_iteration(int phase, bool with_events) {
    if (with_events) HIPCHECK(hipEventRecord(ev[8], st));
    // MLP_DEBUG_SYNC=1 with MLP_NO_GRAPH=1 (eager launches): synchronise after every stage and name it — a device fault is then
    // reported at the stage it belongs to
    static const bool debug_sync = std::getenv("MLP_DEBUG_SYNC") != nullptr && !use_graph_env();
    for (int i = 0; i < 6; ++i) {
        launch_stage(phase, kStageOrder[phase][i], with_events);
        if (debug_sync) {
            const hipError_t e = hipStreamSynchronize(st);
            std::fprintf(stderr, "[mlp] phase %d stage %d: %s\n", phase, kStageOrder[phase][i], hipGetErrorString(e));
        }
    }
    if (with_events) HIPCHECK(hipEventRecord(ev[9], st));
}
size_t Engine::nnz_nucleus_cols() {
    size_t s = 0;
    for (int p = 0; p < m_; ++p) {
        const int c = col_nnz(h_basic_vars[p]);
        if (c != 1) s += (size_t)c;
    }
    return s;
}

// ------------------------------------------------------------------ engine-level stepping
// The same kernels, one stage per call, paced by the host: what a host-side Solver (the reference's
// choose_pivot / pivot split, solver.rs:487-547) would call through the C ABI.  step_open() takes the
// pricing decision for the phase initial_solve would run next; step_stage() must then be called with
// the stages in the order of kStageOrder; after STAGE_APPLY the basis bookkeeping has been replayed
// and the next iteration is already priced.
void Engine::fill_step_info(StepInfo* out, int phase) const {
    if (!out) return;
    const IterState& it = h_ctl->it;
    out->status = it.status; out->phase = phase;
    out->col = it.q; out->row = it.r;
    out->entering_var = it.entering_var; out->leaving_var = it.leaving_var;
    out->pivot_coeff = it.pivot_coeff; out->step = it.entering_diff; out->objective = it.obj;
    out->nucleus_size = (uint64_t)k_;
    out->next_stage = (step_pos >= 0 && step_pos < 6) ? kStageOrder[phase][step_pos] : -1;
}
int Engine::step_finish(int phase, int status) {  // terminal statuses: what optimize / restore_feasibility do
    step_pos = -1;
    if (phase == 0 && status == ITER_OPTIMAL) {
        dual_feasible = true;
        resume_in_optimize = false;
        enable_pse = false;  // solver.rs:482
    } else if (phase == 1 && status == ITER_FEASIBLE) {
        primal_feasible = true;
    }
    return status;
}
int Engine::step_open(StepInfo* out) {
    if (shard_world > 1) throw MlpError(-1, "the stepping API is not available on a sharded solution");
    int phase;
    if (!primal_feasible) phase = 1;
    else if (!dual_feasible) {
        if (!resume_in_optimize) recalc_obj_coeffs();
        resume_in_optimize = true;
        phase = 0;
    } else {
        pull_ctl();
        h_ctl->it.status = ITER_OPTIMAL;
        step_pos = -1;
        fill_step_info(out, 0);
        return ITER_OPTIMAL;
    }
    step_phase = phase;
    if (str_now) {  // stepped iterations use the dense tableau row (row_coeffs is inspectable in full)
        str_now = false;
        view_dirty = true;
    }
    ensure_beta();  // stepped iterations maintain beta by the recurrence, from an exact base
    pivot_budget = -1;
    budget_exhausted = false;
    sync_view();
    ensure_nucleus_cap(k_ + 2);
    sync_view();
    launch_reset_ring(hview, st);
    launch_clear_work(hview, st);
    if (phase == 0) launch_price_primal(hview, geom(), enable_pse ? 1 : 0, st);
    else launch_price_dual(hview, geom(), enable_dse ? 1 : 0, st);
    pull_ctl();
    int status = h_ctl->it.status;
    step_pos = 0;
    if (status != ITER_PIVOT) {
        process_records(phase, 1);
        status = step_finish(phase, status);
    }
    fill_step_info(out, phase);
    return status;
}
int Engine::step_stage(int stage, StepInfo* out) {
    if (step_pos < 0 || step_pos >= 6) throw MlpError(-1, "step_stage: no open iteration (call mlp_engine_open first)");
    const int phase = step_phase;
    if (stage != kStageOrder[phase][step_pos])
        throw MlpError(-1, "step_stage: stages must follow the iteration's order (next expected: " +
                               std::to_string(kStageOrder[phase][step_pos]) + ")");
    if (step_pos == 0) {
        sync_view();
        ensure_nucleus_cap(k_ + 2);
        fac_make_room(1);
        sync_view();
        launch_reset_ring(hview, st);  // one record per stepped iteration
    }
    stepping = true;   // (stepped iterations keep the beta recurrence: every stage's vector is inspectable, tau included)
    batch_lazy = false;
    launch_stage(phase, stage, false);
    stepping = false;
    pull_ctl();
    step_pos += 1;
    int status = h_ctl->it.status;
    if (stage == STAGE_APPLY) {
        // replay the bookkeeping of the iteration just closed; the ring may already hold the terminal
        // record of the NEXT pricing decision (optimal / feasible), which process_records returns
        int res = process_records(phase, 1);
        if (h_ctl->max_pivot_err > stats.max_pivot_err) stats.max_pivot_err = h_ctl->max_pivot_err;
        step_pos = 0;
        status = res;
        if (res != ITER_PIVOT) status = step_finish(phase, res);
        else if (!(h_ctl->max_pivot_err <= refresh_tol) && (k_ > 0 || fac_on_)) rebuild_inverse();
        h_ctl->it.status = status;
    } else if (status != ITER_PIVOT && status != ITER_FLIP) {
        // UNBOUNDED / INFEASIBLE / SINGULAR decided inside the iteration
        process_records(phase, 1);
        status = step_finish(phase, status);
    }
    fill_step_info(out, phase);
    return status;
}

hipGraphExec_t Engine::get_graph(int phase, int multi) {
    const int pse = enable_pse ? 1 : 0;
    const Geom g = geom();
    hipGraphExec_t& ge = gexec[phase][pse][multi];
    hipGraph_t& gg = ggraph[phase][pse][multi];
    if (ge && std::memcmp(&ggeom[phase][pse][multi], &g, sizeof(Geom)) == 0) return ge;
    if (ge) {
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(gg);
        ge = nullptr;
        gg = nullptr;
    }
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipStreamSynchronize(st2));
    HIPCHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    try {
        const int iters = multi ? graph_iters : 1;
        for (int i = 0; i < iters; ++i) record_iteration(phase, false);
    } catch (...) {
        hipGraph_t junk = nullptr;
        (void)hipStreamEndCapture(st, &junk);
        if (junk) (void)hipGraphDestroy(junk);
        throw;
    }
    HIPCHECK(hipStreamEndCapture(st, &gg));
    HIPCHECK(hipGraphInstantiate(&ge, gg, nullptr, nullptr, 0));
    ggeom[phase][pse][multi] = g;
    return ge;
}

// Consume the per-pivot records of a batch: statistics, trace and the host mirror of the basis
// bookkeeping (solver.rs:1088-1091).  Returns the terminal status or ITER_PIVOT to continue.
int Engine::process_records(int phase, int launched) {
    (void)launched;
    int n = std::min(h_ctl->ring_n, RING);
    int result = ITER_PIVOT;
    for (int i = 0; i < n; ++i) {
        const PivotRec& r = h_ctl->ring[i];
        if (pivot_budget == 0) break;  // a decision taken beyond the budget (next-iteration pricing) does not count
        if (pivot_budget > 0) pivot_budget -= 1;
        if (r.status == ITER_PIVOT || r.status == ITER_FLIP) {
            stats.iterations += 1;
            lifetime_pivots += 1;
            iters_since_recalc += 1;
            iters_since_polish += 1;
            if (r.phase == 0) {
                stats.primal_iters += 1;
                if (r.status == ITER_PIVOT && batch_lazy) beta_stale = true;  // that pivot skipped the beta recurrence
            } else {
                stats.dual_iters += 1;
            }
            values_dirty = true;
            if (r.status == ITER_FLIP) {
                stats.bound_flips += 1;
                if (trace) trace_log.push_back({r.phase, r.q, -1, r.entering_var, -1, 0.0, r.obj});
            } else {
                int ev_ = r.entering_var, lv = r.leaving_var;
                h_basic_vars[r.r] = ev_;
                h_var_loc[ev_] = r.r;
                h_nb_vars[r.q] = lv;
                h_var_loc[lv] = -1 - r.q;
                nnz_nonbasic += (size_t)col_nnz(lv);
                nnz_nonbasic -= (size_t)col_nnz(ev_);
                stats.basis_changes += 1;
                if (r.kase >= 0 && r.kase < 5) stats.kase[r.kase] += 1;
                if (trace) trace_log.push_back({r.phase, r.q, r.r, ev_, lv, r.pivot_coeff, r.obj});
            }
        } else {
            result = r.status;
        }
    }
    (void)phase;
    return result;
}

int Engine::run_loop(int phase) {
    if (phase == 1) ensure_beta();  // the dual pricing reads beta
    if (fac_on_) pull_ctl();        // (the count of pending rank-1 terms)
    str_clean = false;  // (whatever ran since the last loop may have written alpha_r / helper densely)
    batch_lazy = lazy_now(phase);
    bool fpk_fresh = false;  // (this loop has built its packed copy of the nucleus columns)
    for (;;) {
        if (pivot_budget == 0) {
            budget_exhausted = true;
            return ITER_PIVOT;
        }
        // profile mode: every 8th batch is ONE eager iteration bracketed by HIP events on the
        // launch stream (sweep and fused pass), the others are graph replays as usual
        const bool long_run = use_graph && graph_iters > 1 && graph_batches_in_geom >= 8;
        const bool sample = profile && (sample_every == 1 || batches_run % (long_run ? 4 : 8) == 0);
        batches_run += 1;
        // Capturing a graph costs milliseconds and is invalidated by every add_constraint (m changes),
        // so short warm-start re-solves run eagerly; the graph is captured once the same geometry has
        // survived a few iterations.
        if (!hview.nb_order && use_order && lifetime_pivots >= order_from) view_dirty = true;  // time to switch the order on
        sync_view();
        if (hview.nb_order) refresh_nb_order(false);  // every `order_every` pivots (same buffer: captured graphs stay valid)
        if (hview.nb_order && use_pack && pack_built != (hview.pk_ptr != nullptr)) {  // the packed copy came or went
            view_dirty = true;
            sync_view();
        }
        // pulled F product (fpull.inc): the packed copy of the nucleus columns is rebuilt at the first batch of this loop and every
        // fpk_every_ pivots (the entries of columns that have left the basis since are dead weight in the rows: dropped by the rebuild)
        if (phase == 0 && fpk_wanted() && hview.pb_on && (!hview.pb_det || shard_world > 1) && (!fpk_fresh || !fpk_valid_ || lifetime_pivots - fpk_built_at_ >= (uint64_t)fpk_every_)) {
            fpk_rebuild();
            fpk_fresh = true;
        }
        // pivots the multi-kernel path still has to take before the hypersparse kernel is tried again (after a bail-out)
        int hyper_gap = 0;
        if (hyper_off_until > lifetime_pivots && hyper_capable(phase))
            hyper_gap = (int)std::min<uint64_t>(hyper_off_until - lifetime_pivots, (uint64_t)1 << 20);
        if (!sample && hyper_gap == 0 && hyper_capable(phase)) {
            // Hypersparse iteration (hyper.inc): up to RING dual iterations in ONE launch of one workgroup doing only
            // support-restricted work.  Same records, same device state: the two paths alternate freely.
            int B = RING;
            if (pivot_budget > 0 && (int64_t)B > pivot_budget) B = (int)pivot_budget;
            ensure_nucleus_cap(k_ + B + 1);
            if (fac_on_) continue;  // (the capacity request switched to the compact factor: plan the batch again)
            ensure_hyper();
            sync_view();
            launch_reset_ring(hview, st);
            launch_clear_work(hview, st);
            launch_hyper_dual(hview, enable_dse ? 1 : 0, B, hyper_heavy, st);
            str_clean = false;  // (the kernel writes alpha_r on its touched columns without zeroing them)
            pull_ctl();
            const uint64_t before = stats.iterations;
            int res = process_records(phase, B);
            stats.hyper_iters += stats.iterations - before;
            if (h_ctl->hyper_bail) {
                // the kernel declined an iteration (a list did not fit its LDS capacities, or the update was too much work for
                // one workgroup): nothing of it was applied; the multi-kernel path takes over for a while — the longer, the
                // more often this happens in a row (a model that is not hypersparse settles on the multi-kernel path)
                stats.hyper_bails += 1;
                if (h_ctl->hyper_bail > 0 && h_ctl->hyper_bail < 10) stats.hyper_bail_reason[h_ctl->hyper_bail] += 1;
                // Dense iterations come in clusters (measured on config 3: handing over one pivot at a time costs 173 bail-outs
                // and 185 ms against 28 and 172 ms with a back-off): after a run of >= 16 hypersparse iterations the multi-kernel
                // path takes that one pivot only; after a short run it keeps going for 4, 8, ... 512 pivots before the next attempt
                if (stats.iterations - before >= 16) {
                    hyper_bail_streak = 0;
                    hyper_off_until = lifetime_pivots + 1;
                } else {
                    hyper_bail_streak = std::min(hyper_bail_streak + 1, hyper_backoff_max);
                    hyper_off_until = lifetime_pivots + ((uint64_t)2 << hyper_bail_streak);
                }
            } else if (stats.iterations - before >= 16) {
                hyper_bail_streak = 0;
            }
            if (h_ctl->max_pivot_err > stats.max_pivot_err) stats.max_pivot_err = h_ctl->max_pivot_err;
            if (res == ITER_PIVOT && !(h_ctl->max_pivot_err <= refresh_tol) && (k_ > 0 || fac_on_)) rebuild_inverse();
            if (res != ITER_PIVOT) return res;
            continue;
        }
        // Sparse tableau row (small nucleus, one GPU): decided per batch — the geometry is baked into the graphs.  alpha_r /
        // helper must be zero outside the touched entries when such a batch starts: any dense sweep (or the hypersparse
        // kernel) in between leaves them dirty.
        {
            // (sharded solves too: a rank lists and pulls the touched columns of its own block of non-basic positions)
            const bool want = str_kmax > 0 && !stepping && !fac_on_ && k_ + RING + 1 <= str_kmax && max_row_nnz_ <= 4096;
            if (want != str_now) {
                str_now = want;
                view_dirty = true;
            }
            // (one workgroup walks all of W in k_small_basis: it pays while the nucleus is small — per pivot of config 4 from the
            // slack basis 55-57 us against 60-62 up to k = 72, 67.0 against 68.8 at k = 104, 78.9 against 77.9 at k = 120, 105
            // against 98 at k = 135 (tools/small_basis_curve.py) — so the form is chosen per batch by the size the nucleus can reach)
            sb_now = want && k_ + RING + 1 <= sb_kmax;
            if (shard_world > 1 && !shard_live_ && !(shard_defer_ && want && !pump_backend_)) {
                // the column-block sharding goes live (every rank takes this decision at the same pivot: k_ is replicated) — after the
                // ranks have shown each other that they ARE the same replica (golive_check throws on every rank otherwise)
                if (shard_defer_) golive_check();
                shard_live_ = true;
                view_dirty = true;
            }
            if (str_now) {
                ensure_hyper();  // the epoch stamps
                d_str_list.ensure((size_t)num_vars + (size_t)m_ + 192, 0, st);  // touched columns | positions of supp(alpha_q)
                if (!hview.str_list || hview.str_list != d_str_list.p) view_dirty = true;
                sync_view();
                if (!str_clean) {
                    HIPCHECK(hipMemsetAsync(d_alpha_r.p, 0, sizeof(double) * (size_t)num_vars, st));
                    HIPCHECK(hipMemsetAsync(d_helper.p, 0, sizeof(double) * (size_t)num_vars, st));
                    launch_str_reset(hview, st);  // new stamp epoch, aq_n = str_n = 0
                    str_clean = true;
                }
            } else {
                str_clean = false;
            }
        }
        const bool have_graph = gexec[phase][enable_pse ? 1 : 0][0] != nullptr;
        // (the one or two iterations handed over by the hypersparse kernel run eagerly: no graph is captured for them)
        const bool brief = hyper_gap > 0 && hyper_gap <= 4;
        // (pump transports launch eagerly: measured on the test box, a replayed graph keeps the pump's second stream from making
        // progress until the graph has drained — its kernels then wait for records that cannot arrive)
        const bool graph_now = use_graph && !pump_backend_ && !sample && !brief && (have_graph || eager_iters_in_geom >= 4);
        if (!have_graph && !brief) eager_iters_in_geom += 1;
        // long runs move on to graphs of several iterations and batches of a full record ring: one
        // launch and one host round trip cover more pivots (6 990 -> 7 300 pivots/s on config 4); short
        // warm-start re-solves never pay for capturing the longer graph
        // (round 5: a COLD solve — from the slack basis or a loaded basis, not a warm-start re-solve — captures the graph of several iterations
        // together with the first one-iteration graph: between two graph launches the device idles ~8 us (in-kernel timeline: 33 us of
        // kernels per pivot, 43 us per pivot on the clock), between two kernels of one graph ~1.4)
        // (second session of round 5: the compact factor too — a batch between two refactorisations is 32 or 64 pivots, three to six graphs
        // of ten iterations instead of one graph launch, and ~8 us of idle device, per pivot)
        const bool fac_multi = true;
        const bool multi = graph_now && (long_run || cold_start_) && graph_iters > 1 && (!fac_on_ || fac_multi);
        int B = graph_now ? (multi ? RING : batch) : 1;
        if (pivot_budget > 0 && (int64_t)B > pivot_budget) B = (int)pivot_budget;
        if (hyper_gap > 0 && B > hyper_gap) B = hyper_gap;
        bool use_multi = multi;  // whole multi-iteration graphs first, the remainder of the batch as one-iteration graphs
        const bool graph_batch = graph_now;
        {   // small-nucleus primal head: the batch must end before the nucleus can outgrow the kernel's slots
            const int kmax = std::min(ph_kmax >= 0 ? ph_kmax : primal_head_kmax(std::max(max_col_nnz_, 1)), primal_head_kmax(std::max(max_col_nnz_, 1)));
            bool ph = str_now && phase == 0 && enable_pse && batch_lazy && shard_world == 1 && !stepping && !fac_on_ && kmax > 0 &&
                      max_row_nnz_ <= HEAD_LIST_CAP && !hview.pb_on && !hview.det_pull && hview.rowinfo != nullptr;
            if (ph) {
                const int room = kmax - k_;  // every pivot may add one slot
                const int Bp = std::min(B, room);
                if (Bp >= std::min(B, 4)) B = Bp;
                else ph = false;
            }
            ph_now = ph;
        }
        {
            const bool fac_before = fac_on_;
            ensure_nucleus_cap(k_ + B + 1);  // (may switch to the compact factor: then the batch is planned again)
            if (fac_on_ != fac_before) continue;
        }
        if (fac_on_) {
            // compact factor: every basis change of the batch appends a rank-1 term; refactor (re-peel the current basis)
            // when fewer than a batch of them fit — the reference's rule is eta nnz >= LU nnz (solver.rs:1096-1103)
            int room = fac_period_ - h_ctl->nlow;
            if (room < std::min(fac_period_, std::max(1, std::min(B, 8)))) {
                if (!fac_refactor()) {  // the basis no longer peels: back to the explicit inverse
                    fac_leave();
                    continue;
                }
                room = fac_period_;
            }
            if (B > room) B = room;
        }
        sync_view();
        const DevView& dv = hview;
        launch_reset_ring(dv, st);
        launch_clear_work(dv, st);
        if (phase == 0) launch_price_primal(dv, geom(), enable_pse ? 1 : 0, st);  // opens the first iteration
        else launch_price_dual(dv, geom(), enable_dse ? 1 : 0, st);
        if (graph_batch) {
            const int nfull = use_multi ? B / graph_iters : 0;
            hipGraphExec_t gm = (use_multi && (nfull > 0 || cold_start_)) ? get_graph(phase, 1) : nullptr;  // (cold solve: captured ahead of its first use)
            hipGraphExec_t g1 = (B - nfull * graph_iters > 0 || !gm) ? get_graph(phase, 0) : nullptr;
            for (int i = 0; i < nfull; ++i) HIPCHECK(hipGraphLaunch(gm, st));
            for (int i = nfull * graph_iters; i < B; ++i) HIPCHECK(hipGraphLaunch(g1, st));
            graph_batches_in_geom += 1;
        } else {
            if (sample)
                for (auto& e : ev)
                    if (!e) HIPCHECK(hipEventCreate(&e));
            record_iteration(phase, sample);
        }
        const size_t nnz_before = nnz_nonbasic;
        const int k_before = k_;
        const size_t nnz_nuc_before = sample ? nnz_nucleus_cols() : 0;
        pump_until_idle();  // (pump transports: deliver the exchanges of the batch while it runs)
        pull_ctl();
        int res = process_records(phase, B);
        if (sample && h_ctl->ring_n >= 1 && h_ctl->ring[0].status == ITER_PIVOT) {
            float ms = 0.f;
            if (geom().str) {
                if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) {
                    stats.str_ms += ms;
                    stats.str_launches += 1;
                }
            } else if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) {
                stats.sweep_ms += ms;
                const double sh = shard_world > 1 ? 1.0 / shard_world : 1.0;  // a rank sweeps its own column block
                stats.sweep_bytes += sh * (12.0 * (double)nnz_before + 16.0 * num_vars) + 16.0 * m_;
                stats.sweep_launches += 1;
            }
            // (a folding pivot whose fold produced the v partials itself skipped the streaming pass: its empty launch is no sample)
            const bool pass_skipped = hview.lrJ > 0 && h_ctl->fold && geom().big &&
                                      fold_fuses_v(hview, enable_pse ? 1 : 0, lazy_now(phase) ? 0 : 1, 0);
            if (k_before > 1 && !pass_skipped && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) {
                // the pass over the nucleus inverse, stamped by the kernel itself: in place it reads and writes W (16 k^2),
                // in the delayed-update mode it only reads W0 (8 k^2; a rank of the row-sharded pass reads its strips only)
                stats.fused_ms += ms;
                const double stream_share = hview.wshard ? 1.0 / shard_world : 1.0;
                stats.fused_bytes += (hview.lrJ > 0 ? 8.0 * stream_share : 16.0) * (double)k_before * (double)k_before;
                stats.fused_launches += 1;
            }
            {
                float fms = 0.f;
                if (k_before > 1 && hview.lrJ > 0 && h_ctl->fold && geom().big && hipEventElapsedTime(&fms, ev[10], ev[11]) == hipSuccess) {
                    stats.fold_ms += fms;  // this pivot folded first: read + write of W0 (and, fused, its part of v)
                    stats.fold_bytes += 16.0 * (double)k_before * (double)k_before;
                    stats.fold_launches += 1;
                }
            }
            if (hipEventElapsedTime(&ms, ev[4], ev[5]) == hipSuccess) {
                stats.update_ms += ms;
                stats.update_launches += 1;
            }
            if (hipEventElapsedTime(&ms, ev[6], ev[7]) == hipSuccess) {
                // FTRAN of the entering column (as built): the listed columns of W (8 k |list|),
                // the column itself and the F push through the nucleus columns (12 B per entry)
                stats.ftran_ms += ms;
                stats.ftran_bytes += 8.0 * (double)k_before * (double)h_ctl->ring[0].klist_n + 12.0 * (double)nnz_nuc_before +
                                     12.0 * (double)col_nnz(h_ctl->ring[0].entering_var);
                stats.ftran_launches += 1;
            }
            if (hipEventElapsedTime(&ms, ev[8], ev[9]) == hipSuccess) {
                stats.iter_ms += ms;
                stats.iter_samples += 1;
            }
            (void)hipGetLastError();  // an event pair that was not recorded this iteration is not an error
        }
        // drift monitor: the pivot element from FTRAN and from the tableau row must agree
        if (h_ctl->max_pivot_err > stats.max_pivot_err) stats.max_pivot_err = h_ctl->max_pivot_err;
        if (res == ITER_PIVOT && !(h_ctl->max_pivot_err <= refresh_tol) && (k_ > 0 || fac_on_)) rebuild_inverse();
        if (res == ITER_STALL && !ratio_two && shard_world == 1 && !(phase == 1 && h_ctl->forced)) {
            // The in-kernel wait between the two Harris passes timed out: the grid was not co-resident (another process
            // holds the CUs, or the occupancy estimate was wrong for this partition).  Nothing of the iteration has been
            // applied — the stalled kernel only halted the batch — so switch to the two-launch form for good (the
            // geometry changes: graphs are re-captured) and run the iteration again: the next batch clears the work
            // vectors and re-takes the same pricing decision from the unchanged d / gamma / x_B / beta.
            ratio_two = true;
            stats.ratio_stalls += 1;
            str_clean = false;  // the stalled iteration stamped / listed columns without closing: the next batch re-zeroes and clears the counters
            if (pivot_budget >= 0) pivot_budget += 1;  // the stalled record consumed one unit
            HIPCHECK(hipMemsetAsync(d_ticket.p, 0, sizeof(unsigned) * d_ticket.cap, st));
            continue;
        }
        if (res != ITER_PIVOT) return res;
    }
}

void Engine::ensure_beta() {
    if (!beta_stale) return;
    flush_lowrank();  // W0 must be the whole inverse
    sync_view();
    launch_exact_beta(hview, st);
    HIPCHECK(hipStreamSynchronize(st));
    beta_stale = false;
    stats.beta_rebuilds += 1;
}

// Hypersparse iteration (hyper.inc): the dual loop without primal steepest edge, one GPU, in-place nucleus inverse, every
// row / column short enough for the in-kernel stage heads.  auto: models with few non-zeros per row (the regime where the
// multi-kernel iteration loses to reach-restricted CPU work); MLP_HYPER=1 / 0 force it on (where it applies) / off.
bool Engine::hyper_capable(int phase) const {
    if (hyper_mode == 0 || phase != 1 || enable_pse || shard_world != 1 || stepping || fac_on_) return false;
    if ((lr_force >= 0 ? lr_force : (cap_ >= 8192 ? 32 : 0)) != 0 || use_banded()) return false;
    if (cap_ > 4096) return false;  // (the kernel's dense passes over the nucleus slots hold HY_KU = 8 elements per thread: 4 096 slots)
    // the launch reserves capacity for k_ + RING + 1 slots first (ensure_nucleus_cap): that must not double a 4 096-slot W
    if (std::min(k_ + RING + 1, std::max(m_, 1)) > 4096) return false;
    if (max_col_nnz_ > HEAD_LIST_CAP || max_row_nnz_ > HEAD_LIST_CAP) return false;
    if (hyper_mode == 1) return true;
    // few non-zeros per row OR per column (the TSP relaxations of config 5: 129 per degree row, 2 + cuts per column)
    return m_ > 0 && (double)h_rcol.size() <= 16.0 * (double)std::max(m_, num_vars);
}
void Engine::ensure_hyper() {
    const size_t need = (size_t)num_vars + (size_t)m_;
    if (d_hy_stamp.p && hy_stamp_len == need) return;
    HIPCHECK(hipStreamSynchronize(st));
    // [stamps: n + m | variable -> slot: n + m | (8-byte aligned) column ranges: 2 m] ints, scores: m doubles
    d_hy_stamp.ensure(2 * need + 2 + 2 * (size_t)m_ + 64, 0, st);
    d_hy_score.ensure((size_t)m_ + 8, 0, st);
    HIPCHECK(hipMemsetAsync(d_hy_stamp.p, 0, sizeof(int) * d_hy_stamp.cap, st));  // the epoch only grows: zero is "never"
    hy_stamp_len = need;
    view_dirty = true;
}

// ------------------------------------------------------------------ compact factor of the basis (factor.inc)
constexpr int FAC_MAX_LEVELS = 4096;
void Engine::fac_alloc() {
    // workgroup j of k_fac_solve reduces the coefficient of pending term j: never more terms than workgroups (a partitioned device —
    // CPX mode, ~32 CUs — with MLP_FACTOR_J=64 would silently drop the terms beyond the grid: ADVICE r4)
    fac_J_ = std::max(1, std::min(fac_J_, fac_solve_grid_blocks()));
    fac_period_ = std::min(fac_period_, fac_J_);
    const size_t mm = (size_t)std::max(m_, 1), NN = (size_t)std::max(N_, 1), J = (size_t)fac_J_;
    d_fac_pos_of_var.ensure(NN, 0, st); d_fac_var_of_pos.ensure(mm, 0, st); d_fac_prow.ensure(mm, 0, st);
    d_fac_items.ensure(mm, 0, st); d_fac_lptr.ensure(FAC_MAX_LEVELS + 2, 0, st); d_fac_meta.ensure(16, 0, st);
    d_fac_tmp.ensure(8 * mm, 0, st); d_fac_counters.ensure(4, 0, st);
    {   // the work vectors of the solves are zero outside a solve (the levels a solve skips read as zero)
        const double* before = d_fac_x0.p;
        d_fac_x0.ensure(2 * mm, 0, st);
        if (d_fac_x0.p != before) HIPCHECK(hipMemsetAsync(d_fac_x0.p, 0, sizeof(double) * d_fac_x0.cap, st));
    }
    d_fac_pval.ensure(mm, 0, st); d_fac_coef.ensure(2 * 64 + 2, 0, st); d_fac_part.ensure((size_t)3 * 64 * 1024, 0, st);
    d_fac_U.ensure(J * mm, 0, st); d_fac_V.ensure(J * mm, 0, st);
    d_fac_bpos.ensure(std::max(FAC_BMAX, FAC_SB_MAX), 0, st); d_fac_brow.ensure(std::max(FAC_BMAX, FAC_SB_MAX), 0, st);
    {   // resolved edge lists: at most the entries of A (incl. the slack identity) on either side
        const size_t nz = h_rcol.size() + 8;
        d_fac_irow.ensure(mm, 0, st); d_fac_ipiv.ensure(mm, 0, st);
        d_fac_fptr.ensure(mm + 2, 0, st); d_fac_bptr.ensure(mm + 2, 0, st);
        d_fac_fidx.ensure(nz, 0, st); d_fac_fval.ensure(nz, 0, st); d_fac_bidx.ensure(nz, 0, st); d_fac_bval.ensure(nz, 0, st);
        d_fac_eslot.ensure(2 * nz, 0, st);
        d_scan_tmp.ensure((mm + 2) / 4096 + 8, 0, st);
        d_fac_lev3.ensure(5 * mm, 0, st);  // (+ 2 m: place of a position / of a row's pivot position in the item list)
        d_fac_tprog.ensure(2 * mm, 0, st);  // (records of the small levels' items, both directions)
    }
    {   // (no bump yet: every row is a pivot row of the peel)
        const int* before = d_fac_bslot_of_row.p;
        d_fac_bslot_of_row.ensure(mm, 0, st);
        if (d_fac_bslot_of_row.p != before) HIPCHECK(hipMemsetAsync(d_fac_bslot_of_row.p, 0xFF, sizeof(int) * d_fac_bslot_of_row.cap, st));
    }
    {   // the right-hand side vector is zero outside a solve, the barrier words outside a kernel
        const double* before = d_fac_rhs.p;
        d_fac_rhs.ensure(mm, 0, st);
        if (d_fac_rhs.p != before) HIPCHECK(hipMemsetAsync(d_fac_rhs.p, 0, sizeof(double) * d_fac_rhs.cap, st));
        const unsigned* b0 = d_fac_bar.p;
        d_fac_bar.ensure(FAC_BAR_FLAG + 64, 0, st);
        if (d_fac_bar.p != b0) HIPCHECK(hipMemsetAsync(d_fac_bar.p, 0, sizeof(unsigned) * d_fac_bar.cap, st));
    }
}
void Engine::fac_fill_view(DevView& v) const {
    v.fac_on = fac_on_ ? 1 : 0;
    v.fac_J = fac_J_;
    v.fac_meta = d_fac_meta.p; v.fac_pos_of_var = d_fac_pos_of_var.p; v.fac_var_of_pos = d_fac_var_of_pos.p;
    v.fac_prow = d_fac_prow.p; v.fac_pval = d_fac_pval.p; v.fac_items = d_fac_items.p; v.fac_lptr = d_fac_lptr.p;
    v.fac_U = d_fac_U.p; v.fac_V = d_fac_V.p; v.fac_rhs = d_fac_rhs.p; v.fac_x0 = d_fac_x0.p; v.fac_coef = d_fac_coef.p; v.fac_part = d_fac_part.p;
    v.fac_bar = d_fac_bar.p;
    v.fac_lev_of_pos = d_fac_lev3.p; v.fac_lev_of_row = d_fac_lev3.p ? d_fac_lev3.p + (size_t)std::max(m_, 1) : nullptr;
    v.fac_reach_of_pos = d_fac_lev3.p ? d_fac_lev3.p + 2 * (size_t)std::max(m_, 1) : nullptr;
    v.fac_skip = fac_skip_ ? 1 : 0; v.fac_flow = fac_flow_ ? 1 : 0;
    v.fac_idx_of_pos = d_fac_lev3.p ? d_fac_lev3.p + 3 * (size_t)std::max(m_, 1) : nullptr;
    v.fac_idx_of_row = d_fac_lev3.p ? d_fac_lev3.p + 4 * (size_t)std::max(m_, 1) : nullptr;
    v.fac_tprog_f = d_fac_tprog.p; v.fac_tprog_b = d_fac_tprog.p ? d_fac_tprog.p + (size_t)std::max(m_, 1) : nullptr;
    v.fac_ltslot = d_fac_ltslot.p; v.fac_segs = d_fac_segs.p; v.fac_WbT = d_fac_WbT.p;
    v.fac_fslot = d_fac_eslot.p; v.fac_bslot = d_fac_eslot.p ? d_fac_eslot.p + h_rcol.size() + 8 : nullptr;
    v.fac_irow = d_fac_irow.p; v.fac_ipiv = d_fac_ipiv.p; v.fac_fptr = d_fac_fptr.p; v.fac_fidx = d_fac_fidx.p; v.fac_fval = d_fac_fval.p;
    v.fac_bptr = d_fac_bptr.p; v.fac_bidx = d_fac_bidx.p; v.fac_bval = d_fac_bval.p;
    v.fac_bpos = d_fac_bpos.p; v.fac_brow = d_fac_brow.p; v.fac_bslot_of_row = d_fac_bslot_of_row.p; v.fac_Wb = d_fac_Wb.p;
    if (d_fac_sb_rec.p) {
        const FacSbWork w = fac_sb_work(std::max(m_, 1));
        v.fac_sb_rec = w.rec; v.fac_sb_lptr = w.lptr; v.fac_sb_oidx = w.oidx; v.fac_sb_oval = w.oval;
        v.fac_sb_trow = w.trow; v.fac_sb_tcol = w.tcol; v.fac_sb_tinv = w.tinv; v.fac_sb_tinvT = w.tinvT;
    } else {
        v.fac_sb_rec = nullptr; v.fac_sb_lptr = nullptr; v.fac_sb_oidx = nullptr; v.fac_sb_oval = nullptr;
        v.fac_sb_trow = nullptr; v.fac_sb_tcol = nullptr; v.fac_sb_tinv = nullptr; v.fac_sb_tinvT = nullptr;
    }
}
// the carve-up of the sparse bump factor's buffers (factor_sb.inc)
FacSbWork Engine::fac_sb_work(int m) const {
    FacSbWork w;
    const size_t B = FAC_SB_MAX;
    int* ip = d_fac_sb_int.p;
    double* dp = d_fac_sb_dbl.p;
    auto it = [&](size_t n) { int* r = ip; ip += n; return r; };
    auto dt = [&](size_t n) { double* r = dp; dp += n; return r; };
    w.rcol = it(B * FAC_SB_RC); w.rcnt = it(B);
    w.crow = it(B * FAC_SB_CC); w.ccnt = it(B);
    w.rstate = it(B); w.cstate = it(B);
    w.lidx = it(B * FAC_SB_LC); w.lcnt = it(B);
    w.pivcol = it(B);
    w.cand_u = it(B); w.cand_cost = it(B); w.won = it(B); w.bid = it(B); w.taken = it(B);
    w.place = it(B);
    w.flags = it(8);
    w.lptr = it(FAC_SB_ROUNDS + 2);
    w.oidx = it(FAC_SB_KINDS * B * FAC_SB_OVS);
    w.trow = it(FAC_SB_DT); w.tcol = it(FAC_SB_DT); w.tidx = it(B);
    w.tK = dt((size_t)FAC_SB_DT * FAC_SB_DT); w.tW = dt((size_t)FAC_SB_DT * FAC_SB_DT);
    w.tinv = dt((size_t)FAC_SB_DT * FAC_SB_DT); w.tinvT = dt((size_t)FAC_SB_DT * FAC_SB_DT);
    w.slot_of_pos = it((size_t)m);  // (last: nothing else moves with the number of rows)
    w.rval = dt(B * FAC_SB_RC); w.lval = dt(B * FAC_SB_LC); w.piv = dt(B); w.oval = dt(FAC_SB_KINDS * B * FAC_SB_OVS);
    w.rec = d_fac_sb_rec.p;
    return w;
}
// The refactorisation (BasisSolver::reset, solver.rs:1286-1303 -> lu_factorize, lu.rs:118-304): an iterated column-singleton
// peel of the CURRENT basis on the device, one level per pair of launches, paced by the host (it reads one counter per level:
// ~20 us per level, amortised over fac_J_ pivots).  Returns false — and leaves the factor as it was — when the peel stops
// before every column is consumed (the basis has a bump: its LU would fill).
bool Engine::fac_refactor(int bump_limit) {
    HIPCHECK(hipStreamSynchronize(st));
    if (m_ <= 0) return false;
    const bool explicit_limit = bump_limit >= 0;
    if (bump_limit < 0) bump_limit = fac_bump_max_;
    fac_alloc();
    sync_view();
    DevView t = hview;  // (also used to probe while the explicit inverse is still the representation in use)
    fac_fill_view(t);
    const size_t mm = (size_t)m_;
    int* cnt = d_fac_tmp.p;
    int* level = cnt + mm;
    int* row_lev = level + mm;
    int* claim = row_lev + mm;
    int* cand_row = claim + mm;
    int* rcnt = cand_row + mm;      // row steps of the peel: active basic columns per row, the bids of the rows, their candidates
    int* claim_r = rcnt + mm;
    int* cand_col = claim_r + mm;
    HIPCHECK(hipMemsetAsync(d_fac_counters.p, 0, 4 * sizeof(int), st));
    launch_fac_peel_init(t, cnt, level, row_lev, claim, rcnt, claim_r, st);
    std::vector<int> lptr(1, 0);
    int total = 0;
    int n_col_steps = 0, n_row_steps = 0, sb_rounds = 0, sb_tail = 0;
    const bool peel_paced = false;  // (the host-paced peel of round 4 remains below as the fallback when a grid barrier of the device peel gives up)
    bool device_peel_done = false;
    if (!peel_paced) {
        // one launch for the whole peel (grid barriers between the phases), one read-back of the level counts
        d_fac_lcount.ensure(FAC_MAX_LEVELS + 2, 0, st);
        launch_fac_peel_all(t, cnt, level, row_lev, claim, cand_row, d_fac_counters.p, d_fac_lcount.p, FAC_MAX_LEVELS, rcnt, claim_r, cand_col, st);
        std::vector<int> lc(FAC_MAX_LEVELS + 2, 0);
        int hc[4] = {0, 0, 0, 0};
        HIPCHECK(hipMemcpyAsync(hc, d_fac_counters.p, sizeof(hc), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(lc.data(), d_fac_lcount.p, sizeof(int) * 64, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (hc[2]) throw MlpError(-2, "singular basis matrix: a basic column lost its last unclaimed row in the peel (solver.rs:1301)");
        if (!hc[1]) {
            const int nl = lc[0];
            if (nl >= 63) {  // (rare: more steps than the first read-back covered)
                HIPCHECK(hipMemcpyAsync(lc.data(), d_fac_lcount.p, sizeof(int) * (size_t)(nl + 1), hipMemcpyDeviceToHost, st));
                HIPCHECK(hipStreamSynchronize(st));
            }
            // sizes of the column steps (in order) and of the row steps (in order); the list of levels = column steps, then the row
            // steps backwards (factor.inc, k_fac_peel_all)
            std::vector<int> csz, rsz;
            int run = 0;
            for (int l = 1; l <= nl; ++l) {
                const int cum = std::abs(lc[l]);
                (lc[l] > 0 ? csz : rsz).push_back(cum - run);
                run = cum;
            }
            total = run;
            n_col_steps = (int)csz.size();
            n_row_steps = (int)rsz.size();
            for (int x : csz) lptr.push_back(lptr.back() + x);
            lptr.push_back(lptr.back());  // the bump: a level of its own in the order, empty in the lists (walked through its dense inverse)
            for (size_t a = rsz.size(); a-- > 0;) lptr.push_back(lptr.back() + rsz[a]);
            device_peel_done = true;
        } else {  // a grid barrier gave up: redo the peel level by level
            HIPCHECK(hipMemsetAsync(d_fac_counters.p, 0, 4 * sizeof(int), st));
            launch_fac_peel_init(t, cnt, level, row_lev, claim, rcnt, claim_r, st);
        }
    }
    for (int lev = 1; !device_peel_done && lev <= FAC_MAX_LEVELS; ++lev) {
        launch_fac_peel_level(t, lev, cnt, level, row_lev, claim, cand_row, d_fac_counters.p, st);
        int hc[4] = {0, 0, 0, 0};
        HIPCHECK(hipMemcpyAsync(hc, d_fac_counters.p, sizeof(hc), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (hc[2]) throw MlpError(-2, "singular basis matrix: a basic column lost its last unclaimed row in the peel (solver.rs:1301)");
        if (hc[0] == total) break;
        total = hc[0];
        lptr.push_back(total);
        n_col_steps += 1;  // (the host-paced fallback runs column steps only: a larger bump, the same order)
        if (total == m_) break;
    }
    if (!device_peel_done) lptr.push_back(lptr.back());  // the (empty) level of the bump
    // What the peel leaves is the BUMP (columns on cycles of the basis graph).  A small bump is carried along with its explicit
    // inverse (Gauss-Jordan here, b^2 doubles); a large one means this basis is not the shape the representation is for.
    const int b = m_ - total;
    // (an explicit limit — the automatic selection's 32 — is meant as given; the default is what either carrier of the bump takes)
    // (a bump that failed the sparse elimination — a row outgrew its slots, or the rounds stalled — is not tried again while it stays that
    // large: every retry is a single-workgroup launch of up to ~1 000 rounds plus a flag read-back, at every refactorisation)
    if (fac_sb_fail_b_ > 0 && (b < fac_sb_fail_b_ - fac_sb_fail_b_ / 8 || fac_sb_fail_skip_ <= 0)) fac_sb_fail_b_ = 0;
    const bool sb_blocked = fac_sb_fail_b_ > 0;
    if (sb_blocked) {
        fac_sb_fail_skip_ -= 1;
        stats.fac_sb_skipped += 1;
    }
    const bool sb_try = b > 0 && b >= fac_sb_from_ && b <= fac_sb_max_ && !sb_blocked;
    if (b > (explicit_limit ? bump_limit : std::max(bump_limit, fac_sb_max_))) return false;
    bool sb_now = false;
    if (b > 0 || fac_bump_ > 0) {
        std::vector<int> hlev(mm), hrow(mm), bslot(mm, -1), bpos, brow;
        HIPCHECK(hipMemcpyAsync(hlev.data(), level, sizeof(int) * mm, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(hrow.data(), row_lev, sizeof(int) * mm, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        for (int p = 0; p < m_; ++p)
            if (hlev[p] == 0) bpos.push_back(p);        // ascending positions / rows: a deterministic slot order
        for (int i = 0; i < m_; ++i)
            if (hrow[i] == 0) {
                bslot[i] = (int)brow.size();
                brow.push_back(i);
            }
        if ((int)bpos.size() != b || (int)brow.size() != b) throw MlpError(-2, "singular basis matrix: the bump of the peel is not square (solver.rs:1301)");
        HIPCHECK(hipMemcpyAsync(d_fac_bslot_of_row.p, bslot.data(), sizeof(int) * mm, hipMemcpyHostToDevice, st));
        if (b > 0) {
            HIPCHECK(hipMemcpyAsync(d_fac_bpos.p, bpos.data(), sizeof(int) * (size_t)b, hipMemcpyHostToDevice, st));
            HIPCHECK(hipMemcpyAsync(d_fac_brow.p, brow.data(), sizeof(int) * (size_t)b, hipMemcpyHostToDevice, st));
        }
        if (sb_try) {
            // the bump as a sparse LU with fill (factor_sb.inc): one launch of one workgroup, two flags back
            const size_t B = FAC_SB_MAX;
            const size_t ni = mm + B * (FAC_SB_RC + FAC_SB_CC + FAC_SB_LC) + 14 * B + 2 * FAC_SB_DT + 8 + FAC_SB_ROUNDS + 2 + FAC_SB_KINDS * B * FAC_SB_OVS + 64;
            const size_t nd = B * (FAC_SB_RC + FAC_SB_LC) + B + FAC_SB_KINDS * B * FAC_SB_OVS + 4 * (size_t)FAC_SB_DT * FAC_SB_DT + 64;
            const int* ib = d_fac_sb_int.p; const double* db = d_fac_sb_dbl.p; const FacSbRec* rb = d_fac_sb_rec.p;
            d_fac_sb_int.ensure(ni, 0, st); d_fac_sb_dbl.ensure(nd, 0, st); d_fac_sb_rec.ensure(FAC_SB_KINDS * B, 0, st);
            if (ib != d_fac_sb_int.p || db != d_fac_sb_dbl.p || rb != d_fac_sb_rec.p) view_dirty = true;
            const FacSbWork w = fac_sb_work(m_);
            t.fac_sb_rec = w.rec; t.fac_sb_lptr = w.lptr; t.fac_sb_oidx = w.oidx; t.fac_sb_oval = w.oval;
            t.fac_sb_trow = w.trow; t.fac_sb_tcol = w.tcol; t.fac_sb_tinv = w.tinv; t.fac_sb_tinvT = w.tinvT;
            launch_fac_sb_factor(t, level, w, b, st);
            int hf[6] = {0, 0, 0, 0, 0, 0};
            HIPCHECK(hipMemcpyAsync(hf, w.flags, sizeof(hf), hipMemcpyDeviceToHost, st));
            HIPCHECK(hipStreamSynchronize(st));
            if (!hf[0] && !hf[1] && hf[3] == 0) {
                sb_now = true;
                sb_rounds = hf[2];
                sb_tail = hf[4];
                stats.fac_sb_factors += 1;
                stats.fac_sb_rounds = (uint64_t)hf[2];
                stats.fac_sb_tail = (uint64_t)hf[4];
            } else {
                stats.fac_sb_fallbacks += 1;  // rows outgrew their slots (or the elimination stalled): this bump is not sparse enough
                fac_sb_fail_b_ = b;           // remembered: no retry until the bump has shrunk by an eighth, or 16 refactorisations have passed
                fac_sb_fail_skip_ = 16;
                if (b > bump_limit) return false;
            }
        }
        if (b > 0 && !sb_now && b > std::min(bump_limit, FAC_BMAX)) return false;  // (the dense inverse holds FAC_BMAX columns)
        if (b > 0 && !sb_now) {
            d_fac_Wb.ensure((size_t)FAC_BMAX * FAC_BMAX, 0, st);
            d_fac_WbT.ensure((size_t)FAC_BMAX * FAC_BMAX, 0, st);
            t.fac_Wb = d_fac_Wb.p;
            t.fac_WbT = d_fac_WbT.p;
            if (hview.fac_Wb != d_fac_Wb.p || hview.fac_WbT != d_fac_WbT.p) view_dirty = true;
            // work arrays of the inversion, kept across refactorisations (grown in steps of 64 rows)
            const size_t brows = (size_t)((b + 63) / 64) * 64;
            d_fac_Kd.ensure(brows * FAC_BMAX, 0, st);
            d_fac_Wtmp.ensure(brows * FAC_BMAX, 0, st);
            d_fac_gjval.ensure(2 * 64 + (size_t)b + 8, 0, st);  // partial maxima of the one-launch form | scratch of the per-column form
            d_fac_gjrow.ensure(2 * 64 + 4, 0, st);               // ... their rows | the two flags
            int* flag = d_fac_gjrow.p + 2 * 64;
            HIPCHECK(hipMemsetAsync(flag, 0, 2 * sizeof(int), st));
            launch_fac_bump_build(t, d_fac_Kd.p, b, st);
            const bool gj_launches = false;  // (the launch-per-column Gauss-Jordan below is the fallback when a grid barrier of the one-launch form gives up)
            int hflag[2] = {0, 0};
            if (!gj_launches) {
                launch_fac_bump_invert(t, d_fac_Kd.p, d_fac_Wtmp.p, d_fac_Wb.p, d_fac_WbT.p, b, flag, d_fac_gjval.p, d_fac_gjrow.p, st);
                HIPCHECK(hipMemcpyAsync(hflag, flag, sizeof(hflag), hipMemcpyDeviceToHost, st));
                HIPCHECK(hipStreamSynchronize(st));
            }
            if (gj_launches || hflag[1]) {  // (a grid barrier of the one-launch form gave up: another process holds the CUs)
                HIPCHECK(hipMemsetAsync(flag, 0, 2 * sizeof(int), st));
                launch_fac_bump_build(t, d_fac_Kd.p, b, st);
                launch_gauss_jordan(d_fac_Kd.p, d_fac_Wb.p, b, FAC_BMAX, flag, d_fac_gjval.p + 2 * 64, st);
                launch_fac_bump_transpose(d_fac_Wb.p, d_fac_WbT.p, b, st);
                HIPCHECK(hipMemcpyAsync(hflag, flag, sizeof(hflag), hipMemcpyDeviceToHost, st));
                HIPCHECK(hipStreamSynchronize(st));
            }
            if (hflag[0]) throw MlpError(-2, "singular basis matrix: the bump of the peel is singular (solver.rs:1301)");
        }
        HIPCHECK(hipStreamSynchronize(st));  // (staged from local vectors)
    }
    fac_bump_ = b;
    stats.fac_bump = (uint64_t)b;
    if ((uint64_t)b > stats.fac_bump_max) stats.fac_bump_max = (uint64_t)b;
    const int nlev = (int)lptr.size() - 1;
    HIPCHECK(hipMemcpyAsync(d_fac_lptr.p, lptr.data(), sizeof(int) * lptr.size(), hipMemcpyHostToDevice, st));
    // the tail: the levels from `tail` on all hold at most FAC_TAIL positions (one workgroup walks them: factor.inc)
    // (the walk of a solve — which levels one workgroup walks alone, their LDS slots, the segments — is planned on the device after the
    // edge lists exist: launch_fac_plan below; meta[3], [6], [7] are its outputs)
    d_fac_ltslot.ensure(2 * ((size_t)FAC_MAX_LEVELS + 2), 0, st);  // LDS slot per level | "has an item with a long edge list" per level
    d_fac_segs.ensure(3 * ((size_t)FAC_MAX_LEVELS + 2), 0, st);
    if (t.fac_ltslot != d_fac_ltslot.p || t.fac_segs != d_fac_segs.p) {
        t.fac_ltslot = d_fac_ltslot.p;
        t.fac_segs = d_fac_segs.p;
        view_dirty = true;
    }
    fac_sb_on_ = sb_now;
    const int meta[16] = {nlev, total, b, nlev, n_col_steps, n_row_steps, 0, 0, sb_now ? 1 : 0, sb_rounds, sb_now ? sb_tail : 0, 0, 0, 0, 0, 0};
    HIPCHECK(hipMemcpyAsync(d_fac_meta.p, meta, sizeof(meta), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)std::max(nlev, 1), st));  // (cnt is free again: the per-level fill cursors)
    launch_fac_peel_fill(t, level, cnt, st);
    // resolved edge lists of the two solves in level order: counts -> exclusive scans -> fill
    HIPCHECK(hipMemsetAsync(d_fac_lev3.p, 0xFF, sizeof(int) * 5 * mm, st));  // (-1: bump positions / rows keep it)
    launch_fac_edges(t, 0, d_fac_fptr.p, d_fac_bptr.p, level, st);
    launch_exclusive_scan(d_fac_fptr.p, d_fac_fptr.p, (long)total + 1, d_scan_tmp.p, st);
    launch_exclusive_scan(d_fac_bptr.p, d_fac_bptr.p, (long)total + 1, d_scan_tmp.p, st);
    launch_fac_edges(t, 1, d_fac_fptr.p, d_fac_bptr.p, level, st);
    launch_fac_plan(t, d_fac_ltslot.p, d_fac_segs.p, st);
    launch_fac_tail_prog(t, d_fac_tprog.p, d_fac_tprog.p + mm, nlev, st);  // the items of the small levels as records, both directions
    {   // the level ranges of the solves (factor.inc): reach of every position, levels in descending order; level of the bump.  Runs of
        // small levels by one workgroup, a level of more than 4 096 positions by the grid
        int hi = nlev - 1;
        for (int lev = nlev - 1; lev >= 0; --lev) {
            const int sz = lptr[lev + 1] - lptr[lev];
            if (sz > 4096) {
                launch_fac_reach_all(t, hi, lev + 1, st);
                launch_fac_reach_level(t, lev, sz, st);
                hi = lev - 1;
            }
        }
        launch_fac_reach_all(t, hi, 0, st);
    }
    {   // level statistics produced before this refactorisation describe the old levels: conservative values until the next producer
        const int stat[4] = {INT_MAX, INT_MAX, 0, INT_MAX};  // fac_aq_hi, fac_rho_hi, fac_aq_lo, fac_aq_reach
        HIPCHECK(hipMemcpyAsync(&d_ctl.p->fac_aq_hi, stat, sizeof(stat), hipMemcpyHostToDevice, st));
        HIPCHECK(hipStreamSynchronize(st));
    }
    HIPCHECK(hipMemsetAsync(&d_ctl.p->nlow, 0, 2 * sizeof(int), st));  // a fresh factor has no pending terms
    HIPCHECK(hipStreamSynchronize(st));  // (lptr / meta were staged from local memory)
    h_ctl->nlow = 0;
    fac_nlev_ = nlev;
    // The refactor period (the reference's rule is eta nnz >= LU nnz, solver.rs:1096-1103).  A pending term costs every dense-rhs solve
    // one more dot product over m; a refactorisation costs the peel (two grid barriers per level) and the factorisation of the bump.
    // Measured: the transport family (20 levels, no bump) 28.4 s with 32 terms, 31.5 s with 64; the config-3 family at 100 000 rows
    // (120-140 levels, bump 1 300-2 200) 25.0 s with 32, 22.9 s with 64.
    // (48, not 32, since the solves got cheaper in round 6: the 200 000-row transport solve 16.81 / 16.11 / 15.88 / 15.90 s at 24 / 32 / 48 / 64)
    if (fac_period_auto_) fac_period_ = std::min(fac_J_, (nlev >= 64 || b >= 64) ? 64 : 48);
    stats.fac_refactors += 1;
    stats.fac_levels = (uint64_t)nlev;
    stats.reinversions += 1;
    return true;
}
// room for `need` more rank-1 terms before an iteration that runs outside run_loop (stepping API, fix_var's forced pivot)
void Engine::fac_make_room(int need) {
    if (!fac_on_) return;
    pull_ctl();
    if (fac_period_ - h_ctl->nlow >= need) return;
    if (!fac_refactor()) fac_leave();
}
bool Engine::fac_enter(int bump_limit) {
    if (fac_on_) return true;
    if (!fac_allowed_under_sharding() || m_ <= 0 || !d_ctl.p) return false;
    ensure_beta();      // (the exact rebuild of the dual edge weights reads the explicit inverse: do it while there is one)
    flush_lowrank();
    if (!fac_refactor(bump_limit)) return false;
    // the explicit inverse and its work arrays go back to the allocator: this representation exists to not hold them
    HIPCHECK(hipStreamSynchronize(st));
    d_W.release(); d_U.release(); d_V.release(); d_Ut.release(); d_part_v.release(); d_part_tau.release();
    cap_ = 0;
    k_ = 0;
    ensure_nucleus_cap(256);  // (small placeholders: the launch geometry of the shared kernels is sized by the capacity)
    HIPCHECK(hipMemcpyAsync(&d_ctl.p->k, &k_, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    fac_on_ = true;
    str_now = false;
    view_dirty = true;
    stats.fac_switches += 1;
    sync_view();
    return true;
}
bool Engine::fac_allowed_under_sharding() const {
    static const bool forced = std::getenv("MLP_FACTOR_SHARED_DEVICE") && std::getenv("MLP_FACTOR_SHARED_DEVICE")[0] == '1';
    return shard_world <= 1 || !ranks_share_device || forced;
}
void Engine::fac_leave() {
    if (!fac_on_) return;
    HIPCHECK(hipStreamSynchronize(st));
    fac_on_ = false;
    view_dirty = true;
    stats.fac_switches += 1;
    const int keep_mode = fac_mode;
    const bool keep_tried = fac_tried_;
    fac_mode = 0;  // (the re-inversion below must not switch straight back)
    try {
        rebuild_inverse();  // classify singleton / nucleus columns, invert the nucleus from A (may end in MLP_ENOMEM on the models this mode exists for)
    } catch (...) {
        fac_mode = keep_mode;
        throw;
    }
    fac_mode = keep_mode;
    fac_tried_ = keep_tried;
    HIPCHECK(hipMemsetAsync(&d_ctl.p->nlow, 0, 2 * sizeof(int), st));
    HIPCHECK(hipStreamSynchronize(st));
}
// One stage of an iteration on the compact factor: the heads, ratio tests, tableau row and update are the kernels of the default
// path (their partition-specific parts are switched off by DevView.fac_on); the solves are k_fac_solve, and the pivot's eta
// transformation is appended as a rank-1 term just before the update kernel zeroes alpha_q and rho behind itself.
void Engine::launch_stage_fac(int phase, int stage, bool with_events) {
    const DevView& dv = hview;
    const Geom g = geom();
    const int pse = enable_pse ? 1 : 0, dse = enable_dse ? 1 : 0;
    const int inl = (dv.banded && phase == 0 && !stepping) ? 1 : 0;
    // A dual iteration without primal steepest edge, run whole (not stepped): the three single-purpose launches around the solves —
    // the leaving row's scalars, the plan after the FTRAN, the new rank-1 term — ride inside the solves (16.8 us of 170 per pivot on the
    // 200 000-row transport instance, profiles/r05c_transport_first60k_kernel_stats.csv); bit-identical (MLP_FACTOR_FUSE=0: A/B, tests)
    static const bool fuse_env = !(std::getenv("MLP_FACTOR_FUSE") && std::getenv("MLP_FACTOR_FUSE")[0] == '0');
    const bool fused = fuse_env && phase == 1 && !pse && dse && fac_pair_ && !stepping;
    static const bool rho_part_env = !(std::getenv("MLP_FACTOR_RHO_PART") && std::getenv("MLP_FACTOR_RHO_PART")[0] == '0');
    switch (stage) {
    case STAGE_FTRAN:
        if (with_events) HIPCHECK(hipEventRecord(ev[6], st));
        if (phase == 0) launch_ftran_prep(dv, 1, st);            // entering column's scalars; the column becomes the right-hand side
        // alpha_q = B^-1 a_q; in a dual iteration rho is known already, so tau = B^-1 rho (solver.rs:1157) shares the walk over the levels
        if (phase == 1 && dse && fac_pair_) launch_fac_solve2(dv, g, 0, 0, 0, 1, 1, st, fused ? 6 : 0);
        else launch_fac_solve(dv, g, 0, 0, 0, nullptr, 0, st);
        if (with_events) HIPCHECK(hipEventRecord(ev[7], st));
        if (phase == 1 && !fused) launch_post_ftran(dv, g, pse, st);       // ||alpha_q||^2 + 1, plan (1 / alpha_q[r])
        break;
    case STAGE_RATIO:
        if (phase == 0) launch_ratio_primal(dv, g, pse, st);
        else launch_ratio_dual(dv, g, st, ar_built_ ? 1 : 0);    // (its finaliser scatters the entering column for the FTRAN)
        break;
    case STAGE_BTRAN:
        if (phase == 1 && !fused) launch_btran_prep(dv, 1, 0, st);         // leaving row's scalars
        // rho = B^-T e_r, ||rho||^2; in a primal iteration alpha_q is known already, so v = B^-T alpha_q (solver.rs:1114) shares the walk
        if (phase == 0 && pse && fac_pair_) launch_fac_solve2(dv, g, 1, 0, 0, 1, 1, st, rho_part_env ? 0 : 8);
        else launch_fac_solve(dv, g, 1, 0, 0, nullptr, 0, st, (fused ? 1 : 0) | (rho_part_env ? 0 : 8));
        break;
    case STAGE_BASIS:
        if (pse && !(phase == 0 && fac_pair_)) launch_fac_solve(dv, g, 1, 1, 1, nullptr, 0, st);   // v = B^-T alpha_q      (solver.rs:1114)
        if (dse && !(phase == 1 && fac_pair_)) launch_fac_solve(dv, g, 0, 1, 1, nullptr, 0, st);   // tau = B^-1 rho        (solver.rs:1157)
        break;
    case STAGE_ROW:
        if (phase == 0) launch_sweep(dv, g, pse ? 1 : 0, 0, st, inl);
        else ar_built_ = launch_sweep(dv, g, 0, 0, st, 0, 1);
        break;
    case STAGE_APPLY:
        if (phase == 1 && pse) launch_sweep(dv, g, 2, 0, st);
        if (with_events) HIPCHECK(hipEventRecord(ev[4], st));
        if (!fused) launch_fac_append(dv, st);
        launch_update_pivot(dv, g, phase, dse, pse, st, inl, 0);
        if (with_events) HIPCHECK(hipEventRecord(ev[5], st));
        break;
    default:
        throw MlpError(-1, "unknown stage");
    }
}

// ------------------------------------------------------------------ loops (solver.rs:470-547)
void Engine::initial_solve() {
    double t0 = now_s();
    budget_exhausted = false;
    if (!primal_feasible) restore_feasibility();
    if (!budget_exhausted && !dual_feasible) {
        if (!resume_in_optimize) recalc_obj_coeffs();  // a budget resume must not recompute d
        resume_in_optimize = true;
        optimize();
    }
    // Polish of a long run: the basic values, too, have been updated incrementally pivot after pivot
    // (solver.rs:1049-1055).  Recompute x_B = B^-1 (b - N x_N) from the basis; if a bound is then violated
    // by more than the tolerance, the dual loop repairs it (the reduced costs were just recomputed and are
    // dual feasible) and the primal loop confirms optimality.  Bounded; short runs never get here.
    for (int round = 0; round < 3 && !budget_exhausted && final_refresh_pivots > 0 &&
                        iters_since_polish >= (uint64_t)final_refresh_pivots; ++round) {
        iters_since_polish = 0;
        if (k_ > 0 || fac_on_) rebuild_inverse();  // fresh K^-1 from A first: the polish is only as accurate as the inverse it uses
        recalc_basic_vals();
        recalc_obj_coeffs();  // objective (and reduced costs) of the recomputed point
        stats.final_refreshes += 1;
        if (basic_values_feasible()) break;
        // The dual loop repairs the bound violations.  The primal loop is deliberately NOT re-entered
        // afterwards: the reference never runs it after a dual phase either (solver.rs:470-485, 633), and
        // on a run that ended in the dual loop the Harris test may have left reduced costs a hair beyond the
        // tolerance, which a primal pricing pass would mistake for an improving (possibly unbounded) column.
        primal_feasible = false;
        restore_feasibility();
    }
    if (!budget_exhausted) {
        resume_in_optimize = false;
        enable_pse = false;  // solver.rs:482
    }
    stats.solve_wall_s += now_s() - t0;
}
// any basic value outside its bounds by more than EPS?  (the dual pricing scan, solver.rs:855-917)
bool Engine::basic_values_feasible() {
    sync_view();
    launch_reset_ring(hview, st);
    launch_price_dual(hview, geom(), enable_dse ? 1 : 0, st);
    pump_until_idle();
    pull_ctl();
    return h_ctl->it.status == ITER_FEASIBLE;
}
// no eligible entering column for the current reduced costs?  (the primal pricing scan, solver.rs:696-739)
bool Engine::reduced_costs_feasible() {
    sync_view();
    launch_reset_ring(hview, st);
    launch_price_primal(hview, geom(), 0, st);
    pump_until_idle();
    pull_ctl();
    const bool ok = h_ctl->it.status == ITER_OPTIMAL;
    launch_reset_ring(hview, st);  // (the scan halts the batch when it finds none)
    return ok;
}
void Engine::optimize() {
    for (;;) {
        int res = run_loop(0);
        if (budget_exhausted) return;
        if (res == ITER_UNBOUNDED) throw LpFail{2};
        if (res == ITER_SINGULAR) throw MlpError(-2, "singular basis (solver.rs:1301)");
        if (res == ITER_COMM) throw MlpError(-3, "sharded pricing: a peer rank did not answer (mailbox spin bound)");
        if (res == ITER_STALL)
            throw MlpError(-3, "primal ratio test: the in-kernel wait timed out (grid not co-resident on this device / partition); "
                               "set MLP_RATIO_TWO_KERNELS=1");
        if (res != ITER_OPTIMAL) throw MlpError(-3, "primal loop ended with unexpected status " + std::to_string(res));
        // Like the reference (solver.rs:1073-1080) the reduced costs are updated incrementally, pivot after
        // pivot.  After a very long run (config 4: 10^6 pivots) they have drifted by ~1e-6, far above the
        // 1e-8 the optimality test works with, so "no eligible column" is re-examined once on reduced costs
        // and an objective recomputed from the basis (solver.rs:1199-1231); the loop resumes if a column
        // turns out to be eligible after all.  Short runs (every parity test) never get here.
        if (final_refresh_pivots <= 0 || iters_since_recalc < (uint64_t)final_refresh_pivots) break;
        recalc_obj_coeffs();
        stats.final_refreshes += 1;
    }
    dual_feasible = true;
}
void Engine::restore_feasibility() {
    int res = run_loop(1);
    if (budget_exhausted) return;
    if (res == ITER_INFEASIBLE) throw LpFail{1};
    if (res == ITER_SINGULAR) throw MlpError(-2, "singular basis (solver.rs:1301)");
    if (res == ITER_COMM) throw MlpError(-3, "sharded pricing: a peer rank did not answer (mailbox spin bound)");
    if (res != ITER_FEASIBLE) throw MlpError(-3, "dual loop ended with unexpected status " + std::to_string(res));
    primal_feasible = true;
}

// ------------------------------------------------------------------ helpers used by the warm-start API
void Engine::calc_col_coeffs(int col) {  // solver.rs:671-677
    sync_view();
    const DevView& dv = hview;
    launch_clear_work(hview, st);
    launch_set_iter(dv, ITER_PIVOT, col, -1, 0.0, 0, st);
    launch_ftran_prep(dv, 0, st);
    if (dv.fac_on) launch_fac_solve(dv, geom(), 0, 0, 0, nullptr, 0, st);
    else launch_ftran_gather(dv, geom(), st);
}
void Engine::calc_row_coeffs(int row, bool with_sweep) {  // solver.rs:680-693
    sync_view();
    const DevView& dv = hview;
    launch_clear_work(hview, st);
    launch_set_iter(dv, ITER_PIVOT, -1, row, 0.0, 0, st);
    launch_btran_prep(dv, 0, 0, st);
    if (dv.fac_on) launch_fac_solve(dv, geom(), 1, 0, 0, nullptr, 0, st);
    else launch_btran(dv, geom(), 0, st);
    if (with_sweep) launch_sweep(dv, geom(), 0, 0, st);
}

// solver.rs:1199-1231.  There is no eta file to flush: W is always current.
void Engine::recalc_obj_coeffs() {
    iters_since_recalc = 0;
    flush_lowrank();  // the dense transposed solve reads W0 as the whole inverse
    sync_view();
    const DevView& dv = hview;
    const Geom g = geom();
    launch_btran_dense(dv, g, st);  // y = B^-T c_B -> rv.y
    launch_recalc_d(dv, g, st);
}

// x_B = B^-1 (b - N x_N) from scratch (solver.rs:1177-1197 is the reference's unused counterpart).
void Engine::recalc_basic_vals() {
    flush_lowrank();  // the dense solve reads W0 as the whole inverse
    // The recomputed x_B must not depend on the order float atomics land in: the ranks of a sharded solve each load the basis
    // in their own process BEFORE sharding is enabled and must start as bit-identical replicas (tools/shard_bitwise.py found
    // 102 basic values differing by 2e-14 between two ranks after the load).  The blocked push takes its deterministic form here.
    struct DetScope {
        Engine* e;
        bool was;
        explicit DetScope(Engine* e_) : e(e_), was(e_->force_det_push_) { e->force_det_push_ = true; e->view_dirty = true; }
        ~DetScope() { e->force_det_push_ = was; e->view_dirty = true; }
    } det_scope(this);
    sync_view();
    DevBuf<double> rhs, r;
    rhs.upload(h_rhs, st);
    r.ensure((size_t)m_ + 8, 0, st);
    // profile mode: the dense-rhs FTRAN of each step (one read of the nucleus inverse: x_K = W r_K) is timed kernel-exactly
    for (int step = 0; step < 3; ++step) {  // the solve, then two steps of iterative refinement with the same inverse: x_B += B^-1 (b - A x)
        if (profile) {
            for (int e : {2, 3})
                if (!ev[e]) HIPCHECK(hipEventCreate(&ev[e]));
            arm_kernel_timing(2, ev[2], ev[3]);
        }
        launch_recalc_basic_vals(hview, geom(), rhs.p, r.p, step > 0 ? 1 : 0, st);
        if (profile) {
            arm_kernel_timing(2, nullptr, nullptr);
            HIPCHECK(hipStreamSynchronize(st));
            float ms = 0.f;
            if (k_ > 1 && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) {
                stats.dense_ftran_ms += ms;
                stats.dense_ftran_bytes += 8.0 * (double)k_ * (double)k_;
                stats.dense_ftran_launches += 1;
            }
            (void)hipGetLastError();
        }
    }
    HIPCHECK(hipStreamSynchronize(st));
    values_dirty = true;
}

void Engine::fix_var(int var, double val) {  // solver.rs:378-415
    cold_start_ = false;  // a warm-start re-solve: short, lazy graph capture
    double t0 = now_s();
    ensure_beta();
    if (val < h_lo[var] || val > h_hi[var]) throw LpFail{1};
    int col;
    if (h_var_loc[var] >= 0) {
        // basic: one forced dual pivot towards `val` (solver.rs:384-391)
        int row = h_var_loc[var];
        ensure_nucleus_cap(k_ + 2);
        fac_make_room(1);
        if (str_now) {  // (the forced iteration runs outside run_loop: dense tableau row, whatever the last batch used)
            str_now = false;
            view_dirty = true;
        }
        sync_view();
        const DevView& dv = hview;
        launch_reset_ring(dv, st);
        launch_clear_work(dv, st);
        launch_set_iter(dv, ITER_PIVOT, -1, row, val, 1, st);
        batch_lazy = false;
        record_iteration(1, false);
        pull_ctl();
        int64_t keep = pivot_budget;
        pivot_budget = -1;
        int res = process_records(1, 1);
        pivot_budget = keep;
        if (res == ITER_INFEASIBLE) throw LpFail{1};
        if (res != ITER_PIVOT) throw MlpError(-2, "fix_var: forced pivot failed");
        col = -1 - h_var_loc[var];
    } else {
        col = -1 - h_var_loc[var];
        calc_col_coeffs(col);
        launch_shift_nonbasic(hview, geom(), col, val, st);
        values_dirty = true;
    }
    uint8_t f = NB_AT_MIN | NB_AT_MAX | NB_FIXED;
    HIPCHECK(hipMemcpyAsync(d_nbflags.p + col, &f, 1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    h_nb_fixed[col] = 1;
    primal_feasible = false;
    budget_exhausted = false;
    restore_feasibility();
    stats.solve_wall_s += now_s() - t0;
}

bool Engine::unfix_var(int var) {  // solver.rs:418-438
    cold_start_ = false;  // a warm-start re-solve: short, lazy graph capture
    if (h_var_loc[var] >= 0) return false;
    int col = -1 - h_var_loc[var];
    if (!h_nb_fixed[col]) return false;
    h_nb_fixed[col] = 0;
    fetch_values();
    double cur = h_xN[col];
    uint8_t f = (uint8_t)((cur == h_lo[var] ? NB_AT_MIN : 0) | (cur == h_hi[var] ? NB_AT_MAX : 0));
    HIPCHECK(hipMemcpyAsync(d_nbflags.p + col, &f, 1, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    dual_feasible = false;
    double t0 = now_s();
    budget_exhausted = false;
    try {
        optimize();
    } catch (LpFail&) {
        throw MlpError(-1, "unfix_var: optimize failed (solver.rs:433 unwrap)");
    }
    stats.solve_wall_s += now_s() - t0;
    return true;
}

void Engine::add_gomory_cut(int var) {  // solver.rs:440-460
    if (h_var_loc[var] < 0) throw MlpError(-1, "add_gomory_cut: variable is not basic (solver.rs:458)");
    int row = h_var_loc[var];
    calc_row_coeffs(row, true);
    std::vector<double> ar(num_vars);
    HIPCHECK(hipMemcpyAsync(ar.data(), d_alpha_r.p, sizeof(double) * num_vars, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    fetch_values();
    std::vector<std::pair<int, double>> terms;
    for (int c = 0; c < num_vars; ++c) {
        double coeff = ar[c];
        double f = std::floor(coeff) - coeff;
        if (f != 0.0) terms.push_back({h_nb_vars[c], f});  // exact zeros carry no information
    }
    std::sort(terms.begin(), terms.end());
    Constraint c;
    c.op = 1;
    c.rhs = std::floor(h_xB[row]) - h_xB[row];
    for (auto& t : terms) {
        c.idx.push_back(t.first);
        c.val.push_back(t.second);
    }
    add_constraint(std::move(c));
}

// Device-side row append (solver.rs:597-613 rebuilds both orientations on the host).  The host has
// already appended the row to its CSR mirror and the variable arrays; here the same row goes to the device copies
// without a pass over the matrix on the host and without re-uploading it: O(row) host work, one O(nnz) device copy.
void Engine::append_row_on_device(const Constraint& c, int slack, int row) {
    const size_t kn = c.idx.size();
    const size_t old_nnz = h_rcol.size() - kn - 1;  // the host mirror already holds the new row (+ slack)
    // --- CSR: append in place (capacity-doubling buffers keep their contents)
    d_rcol.ensure(old_nnz + kn + 1, old_nnz, st);
    d_rval.ensure(old_nnz + kn + 1, old_nnz, st);
    d_rptr.ensure((size_t)m_ + 2, (size_t)m_ + 1, st);
    HIPCHECK(hipMemcpyAsync(d_rcol.p + old_nnz, h_rcol.data() + old_nnz, (kn + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_rval.p + old_nnz, h_rval.data() + old_nnz, (kn + 1) * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_rptr.p + m_ + 1, &h_rptr[m_ + 1], sizeof(int), hipMemcpyHostToDevice, st));
    // --- per-variable arrays: one new element (the slack)
    d_lo.ensure((size_t)N_ + 1, (size_t)N_, st); d_hi.ensure((size_t)N_ + 1, (size_t)N_, st); d_obj.ensure((size_t)N_ + 1, (size_t)N_, st);
    HIPCHECK(hipMemcpyAsync(d_lo.p + N_, &h_lo[N_], sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_hi.p + N_, &h_hi[N_], sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_obj.p + N_, &h_obj[N_], sizeof(double), hipMemcpyHostToDevice, st));
    // --- CSC: one copy kernel into the second buffer set, then swap
    d_row_idx.ensure(kn + 1, 0, st); d_row_val.ensure(kn + 1, 0, st);
    if (kn) {
        HIPCHECK(hipMemcpyAsync(d_row_idx.p, c.idx.data(), kn * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_row_val.p, c.val.data(), kn * sizeof(double), hipMemcpyHostToDevice, st));
    }
    d_cptr_alt.ensure((size_t)N_ + 2, 0, st);
    d_crow_alt.ensure(old_nnz + kn + 1, 0, st);
    d_cval_alt.ensure(old_nnz + kn + 1, 0, st);
    launch_csc_append_row(d_cptr.p, d_crow.p, d_cval.p, N_, row, d_row_idx.p, d_row_val.p, (int)kn, d_cptr_alt.p, d_crow_alt.p,
                          d_cval_alt.p, st);
    HIPCHECK(hipStreamSynchronize(st));  // c.idx / c.val were staged from pageable memory; the old buffers become the spare set
    d_cptr.swap(d_cptr_alt); d_crow.swap(d_crow_alt); d_cval.swap(d_cval_alt);
    // --- host column summaries
    for (size_t p = 0; p < kn; ++p) {
        const int var = c.idx[p];
        h_colnnz[var] += 1;
        if (h_colnnz[var] == 1) {  // a column that was empty so far (tsp.rs:226-235 has such variables) becomes a singleton
            h_single_row[var] = row;
            h_single_val[var] = c.val[p];
        }
        max_col_nnz_ = std::max(max_col_nnz_, h_colnnz[var]);
    }
    max_row_nnz_ = std::max(max_row_nnz_, (int)kn + 1);
    for (size_t p = 0; p < kn; ++p) amax_ = std::max(amax_, std::fabs(c.val[p]));
    amax_ = std::max(amax_, 1.0);  // (the slack entry)
    h_colnnz.push_back(1);
    h_single_row.push_back(row);
    h_single_val.push_back(1.0);
    if (!h_cptr.empty()) {  // the host CSC of the initial build is stale from here on
        std::vector<int>().swap(h_cptr); std::vector<int>().swap(h_crow); std::vector<double>().swap(h_cval);
    }
    colblk_dirty = true;
    banded_dirty = true;
    fpk_valid_ = false;  // (the packed copy of the nucleus columns is a copy of matrix entries)
    view_dirty = true;
}

// solver.rs:549-634.  The new slack is a singleton basic column on the new row, so the nucleus
// inverse is unchanged unless the new row touches a basic singleton column (then: rebuild).
void Engine::add_constraint(Constraint c) {
    cold_start_ = false;  // a warm-start re-solve: short, lazy graph capture
    double t0 = now_s();
    if (!primal_feasible || !dual_feasible) throw MlpError(-1, "add_constraint: model not solved (solver.rs:555-556)");
    // Round 5: the compact factor SURVIVES a new row.  The new slack is a column singleton on the new row — the first level of the peel,
    // whatever else the row holds — so the factor of the extended basis is one more refactorisation (a re-peel of the current basis,
    // ~1 ms: BasisSolver::reset is what the reference does here too, solver.rs:612-613), not a return to the explicit inverse (which
    // the models this representation exists for cannot even allocate).  The pending rank-1 terms are by position / by row of the OLD
    // size: the re-peel starts from the current basis and drops them.
    const bool on_factor = fac_on_;
    ensure_beta();
    if (c.idx.empty()) {
        bool taut = c.op == 0 ? (0.0 == c.rhs) : c.op == 1 ? (0.0 <= c.rhs) : (0.0 >= c.rhs);
        if (taut) return;
        throw LpFail{1};
    }
    fetch_values();
    const int slack = N_, row = m_;
    double smin = c.op == 1 ? 0.0 : c.op == 2 ? -INF : 0.0;
    double smax = c.op == 1 ? INF : 0.0;
    double lhs = 0.0;  // solver.rs:587-595
    bool touches_basic_singleton = false;
    for (size_t p = 0; p < c.idx.size(); ++p) {
        int var = c.idx[p];
        int loc = h_var_loc[var];
        lhs += (loc >= 0 ? h_xB[loc] : h_xN[-1 - loc]) * c.val[p];
        if (loc >= 0 && col_nnz(var) == 1) touches_basic_singleton = true;  // that column stops being a singleton
    }
    double xnew = c.rhs - lhs;
    // matrix: append the CSR row (+ slack), rebuild the CSC (O(nnz), like solver.rs:597-610)
    for (size_t p = 0; p < c.idx.size(); ++p) {
        h_rcol.push_back(c.idx[p]);
        h_rval.push_back(c.val[p]);
    }
    h_rcol.push_back(slack);
    h_rval.push_back(1.0);
    h_rptr.push_back((int)h_rcol.size());
    h_rhs.push_back(c.rhs);
    h_obj.push_back(0.0);
    h_lo.push_back(smin);
    h_hi.push_back(smax);
    HIPCHECK(hipStreamSynchronize(st));
    alloc_row_buffers(m_ + 1);
    append_row_on_device(c, slack, row);  // CSR row appended, CSC re-laid out, derived copies marked for a device rebuild
    m_ += 1;
    N_ += 1;
    ensure_red();
    for (size_t p = 0; p < c.idx.size(); ++p)
        if (h_var_loc[c.idx[p]] < 0) nnz_nonbasic += 1;
    // new basic position `row` holding the slack (singleton on the new row)
    h_basic_vars.push_back(slack);
    h_var_loc.push_back(row);
    int neg1 = -1;
    double one = 1.0, beta0 = 1.0;
    HIPCHECK(hipMemcpyAsync(d_basic_vars.p + row, &slack, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_var_loc.p + slack, &row, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_xB.p + row, &xnew, sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_loB.p + row, &smin, sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_hiB.p + row, &smax, sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_beta.p + row, &beta0, sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_kslot_of_pos.p + row, &neg1, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_srow_of_pos.p + row, &row, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_sdiag_of_pos.p + row, &one, sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_kslot_of_row.p + row, &neg1, sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_pos_of_srow.p + row, &row, sizeof(int), hipMemcpyHostToDevice, st));
    const RowInfo ri_new{1.0, row, -1};
    HIPCHECK(hipMemcpyAsync(d_rowinfo.p + row, &ri_new, sizeof(RowInfo), hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    values_dirty = true;
    view_dirty = true;
    sync_view();
    launch_init_nb_rng(hview, geom(), st);
    if (on_factor) {
        if (!fac_refactor()) fac_leave();  // (the extended basis no longer peels within the bump limit: explicit inverse, as before)
    } else if (touches_basic_singleton) rebuild_inverse();  // a singleton column just gained an entry

    if (enable_pse || enable_dse) {  // solver.rs:615-630: last tableau row feeds the edge norms
        calc_row_coeffs(m_ - 1, enable_pse);
        if (enable_pse) launch_sq_norms_add_row(hview, geom(), st);
        if (enable_dse) launch_copy_rho_sq_to_beta(hview, row, st);
    }
    primal_feasible = false;
    budget_exhausted = false;
    restore_feasibility();
    stats.solve_wall_s += now_s() - t0;
}

// ------------------------------------------------------------------ from-scratch nucleus inverse
// Counterpart of BasisSolver::reset (solver.rs:1286-1303): classify the basic columns (singleton vs
// nucleus), build K = B[R_K, P_K] densely from the CSC and invert it on the device.
void Engine::rebuild_inverse() {
    HIPCHECK(hipStreamSynchronize(st));
    if (fac_on_) {  // compact factor: the counterpart of BasisSolver::reset is a new peel of the current basis
        if (fac_refactor()) return;
        fac_on_ = false;  // the basis no longer peels: fall through to the explicit inverse
        view_dirty = true;
        stats.fac_switches += 1;
    } else if (fac_mode == 1 && fac_allowed_under_sharding() && !stepping) {
        if (fac_enter(32)) return;  // (automatic selection: only a basis that peels almost completely — a bump that grows would flip back)
    }
    HIPCHECK(hipMemsetAsync(&d_ctl.p->nlow, 0, 2 * sizeof(int), st));  // a fresh inverse has no pending terms
    std::vector<int> claimed(m_, -1);
    std::vector<int> nuc_pos;
    h_kslot_of_pos.assign(m_, -1);
    h_srow_of_pos.assign(m_, 0);
    h_sdiag_of_pos.assign(m_, 1.0);
    h_kslot_of_row.assign(m_, -1);
    h_pos_of_srow.assign(m_, 0);
    for (int p = 0; p < m_; ++p) {
        int var = h_basic_vars[p];
        if (col_nnz(var) == 1) {
            int i = h_single_row[var];
            if (claimed[i] >= 0) throw MlpError(-2, "singular basis: two singleton columns on one row");
            claimed[i] = p;
            h_srow_of_pos[p] = i;
            h_sdiag_of_pos[p] = h_single_val[var];
        } else {
            nuc_pos.push_back(p);
        }
    }
    int k = (int)nuc_pos.size();
    k_ = 0;  // nothing to preserve while growing
    ensure_nucleus_cap(std::max(k + (k >= 4096 ? 1024 : 0), 1));  // (a loaded large nucleus gets room for its next thousand columns)
    h_pos_of_kslot.assign(cap_, -1);
    h_row_of_kslot.assign(cap_, -1);
    int s = 0;
    for (int i = 0; i < m_; ++i) {
        if (claimed[i] >= 0) {
            h_pos_of_srow[i] = claimed[i];
        } else {
            if (s >= k) throw MlpError(-2, "singular basis: more uncovered rows than nucleus columns");
            h_kslot_of_row[i] = s;
            h_row_of_kslot[s] = i;
            s += 1;
        }
    }
    if (s != k) throw MlpError(-2, "singular basis: nucleus is not square");
    for (int b = 0; b < k; ++b) {
        h_kslot_of_pos[nuc_pos[b]] = b;
        h_pos_of_kslot[b] = nuc_pos[b];
    }
    k_ = k;
    push_maps();
    if (k > 0) {
        sync_view();
        DevBuf<int> flag;
        flag.ensure(2, 0, st);
        HIPCHECK(hipMemsetAsync(flag.p, 0, 2 * sizeof(int), st));
        int hflag = 0;
        static const bool force_gj = std::getenv("MLP_REINVERT_GJ") != nullptr;  // (debugging: the unblocked Gauss-Jordan at any size)
        if (k >= 384 && !force_gj) {
            // Large nucleus: blocked in-place Gauss-Jordan with partial pivoting (inverse.inc) — K is assembled row-major
            // straight into W and inverted there; the rank-32 update of a block step is the fold kernel of the delayed-update
            // mode.  (Rounds 1-3 called rocSOLVER's dgetrf + dgetri here: the product links no vendor library any more.)
            launch_build_nucleus(hview, geom(), d_W.p, k, st);
            DevBuf<int> piv, src, cidx;
            DevBuf<double> prow, ckey;
            piv.ensure((size_t)k, 0, st);
            src.ensure((size_t)k, 0, st);
            prow.ensure(64, 0, st);
            ckey.ensure((size_t)k / 32 + 8, 0, st);
            cidx.ensure((size_t)k / 32 + 8, 0, st);
            const int nrowbuf = (int)std::min<size_t>(4096, d_part_v.cap / (size_t)ld());
            if (nrowbuf < 1) throw MlpError(-3, "blocked inversion: no row buffer");
            launch_blocked_inverse(hview, k, d_part_v.p, nrowbuf, piv.p, src.p, prow.p, ckey.p, cidx.p, flag.p, st);
            HIPCHECK(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHECK(hipStreamSynchronize(st));
        }
        if (k < 384 || force_gj) {
            // Small nucleus: unblocked Gauss-Jordan with partial pivoting
            DevBuf<double> Kd, scratch;
            Kd.ensure((size_t)k * ld(), 0, st);
            scratch.ensure((size_t)k + 8, 0, st);
            launch_build_nucleus(hview, geom(), Kd.p, k, st);
            launch_gauss_jordan(Kd.p, d_W.p, k, ld(), flag.p, scratch.p, st);
            HIPCHECK(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHECK(hipStreamSynchronize(st));
        }
        if (hflag)
            throw MlpError(-2, "singular basis matrix (solver.rs:1301)");
    }
    stats.reinversions += 1;
}

double Engine::reinvert(bool replace) {
    if (fac_on_) {  // compact factor: a fresh peel of the current basis (there is no incremental inverse to compare with)
        // The return value is NaN — "nothing was compared" — so that a host using reinvert() as a drift probe cannot read a
        // false "no drift" (ADVICE r4); last_reinvert_scale = 0 says the same.
        (void)replace;
        rebuild_inverse();
        last_reinvert_scale = 0.0;
        return std::nan("");
    }
    flush_lowrank();
    pull_maps();
    if (k_ == 0) return 0.0;
    std::vector<int> hp = h_pos_of_kslot, hr = h_row_of_kslot, hkp = h_kslot_of_pos, hkr = h_kslot_of_row,
                     hsr = h_srow_of_pos, hps = h_pos_of_srow;
    std::vector<double> hsd = h_sdiag_of_pos;
    const int kold = k_, capold = cap_, ldold = ld();
    std::vector<double> a((size_t)kold * kold);
    HIPCHECK(hipMemcpy2D(a.data(), (size_t)kold * sizeof(double), d_W.p, (size_t)ldold * sizeof(double),
                         (size_t)kold * sizeof(double), (size_t)kold, hipMemcpyDeviceToHost));
    DevBuf<double> oldW;
    if (!replace) {
        oldW.alloc_exact((size_t)capold * ldold);
        HIPCHECK(hipMemcpy(oldW.p, d_W.p, sizeof(double) * (size_t)capold * ldold, hipMemcpyDeviceToDevice));
    }
    rebuild_inverse();
    double diff = -1.0, scale = 0.0;
    if (k_ == kold) {
        std::vector<double> b((size_t)kold * kold);
        HIPCHECK(hipMemcpy2D(b.data(), (size_t)kold * sizeof(double), d_W.p, (size_t)ld() * sizeof(double),
                             (size_t)kold * sizeof(double), (size_t)kold, hipMemcpyDeviceToHost));
        diff = 0.0;
        for (int s1 = 0; s1 < kold; ++s1)
            for (int s2 = 0; s2 < kold; ++s2) {
                int p = hp[s1], i = hr[s2];
                double x = a[(size_t)s1 * kold + s2];
                double y = b[(size_t)h_kslot_of_pos[p] * kold + h_kslot_of_row[i]];
                double dlt = std::fabs(x - y);
                if (!(dlt <= diff)) diff = dlt;
                if (std::fabs(y) > scale) scale = std::fabs(y);
            }
        last_reinvert_scale = scale;
    }
    if (!replace && cap_ == capold) {  // restore the incremental representation
        h_pos_of_kslot = hp; h_row_of_kslot = hr; h_kslot_of_pos = hkp; h_kslot_of_row = hkr;
        h_srow_of_pos = hsr; h_pos_of_srow = hps; h_sdiag_of_pos = hsd;
        k_ = kold;
        HIPCHECK(hipMemcpy(d_W.p, oldW.p, sizeof(double) * (size_t)capold * ldold, hipMemcpyDeviceToDevice));
        push_maps();
    }
    return diff;
}


// ------------------------------------------------------------------ basis checkpoint
// The reference has no basis I/O; SURVEY §8(d) asks for "pivots from a saved mid-solve basis", which needs
// one.  A checkpoint is a flat, self-describing blob:
//   header | int32 basic_vars[m] | int32 nb_vars[n] | uint8 nb_flags[n] (padded to 8) | f64 x_N[n]
//   mode 1: f32 gamma[n] | f32 beta[m] (padded to 8)        (steepest-edge weights, pricing heuristics)
//   mode 2: f64 x_B[m] | f64 d[n] | f64 gamma[n] | f64 beta[m]
// Loading re-inverts the nucleus from A on the device (the counterpart of BasisSolver::reset,
// solver.rs:1286-1303).  Mode 2 restores every per-pivot vector bit for bit, so the loaded solve continues
// pivot for pivot; modes 0/1 recompute x_B = B^-1 (b - N x_N) (solver.rs:1177-1197) and the reduced costs
// (solver.rs:1199-1231) from the basis and take the weights from the blob (mode 1) or reset them to 1.
namespace {
struct BasisHeader {
    char magic[8];
    uint32_t version, mode;
    uint64_t m, n;
    uint32_t flags;  // bit0 primal_feasible, bit1 dual_feasible, bit2 enable_pse, bit3 enable_dse, bit4 resume_in_optimize
    uint32_t pad;
    double obj;
    uint64_t pivots;
};
constexpr char kBasisMagic[8] = {'M', 'L', 'P', 'B', 'A', 'S', 'I', 'S'};
inline size_t pad8(size_t x) { return (x + 7) & ~(size_t)7; }
size_t basis_blob_size(int mode, size_t m, size_t n) {
    size_t s = sizeof(BasisHeader) + pad8(4 * m) + pad8(4 * n) + pad8(n) + 8 * n;
    if (mode == 1) s += pad8(4 * n) + pad8(4 * m);
    if (mode == 2) s += 8 * (m + n + n + m);
    return s;
}
}  // namespace
std::vector<uint8_t> Engine::save_basis(int mode) {
    if (mode < 0 || mode > 2) throw MlpError(-1, "save_basis: mode must be 0, 1 or 2");
    // a sharded solve keeps reduced costs and weights per column block: what every rank holds in full is the partition
    // itself (basic / non-basic sets, bound flags, x_N) — mode 0
    if (shard_world > 1 && mode != 0) throw MlpError(-1, "save_basis: a sharded solution saves mode 0 only (sets, flags, x_N)");
    HIPCHECK(hipStreamSynchronize(st));
    if (mode >= 1) ensure_beta();
    pull_ctl();
    const size_t mm = (size_t)m_, nn = (size_t)num_vars;
    std::vector<uint8_t> out(basis_blob_size(mode, mm, nn), 0);
    BasisHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, kBasisMagic, 8);
    h.version = 1; h.mode = (uint32_t)mode; h.m = mm; h.n = nn;
    h.flags = (primal_feasible ? 1u : 0u) | (dual_feasible ? 2u : 0u) | (enable_pse ? 4u : 0u) | (enable_dse ? 8u : 0u) |
              (resume_in_optimize ? 16u : 0u);
    h.obj = h_ctl->it.obj;
    h.pivots = stats.iterations;
    uint8_t* p = out.data();
    std::memcpy(p, &h, sizeof(h)); p += sizeof(h);
    std::memcpy(p, h_basic_vars.data(), 4 * mm); p += pad8(4 * mm);
    std::memcpy(p, h_nb_vars.data(), 4 * nn); p += pad8(4 * nn);
    if (nn) HIPCHECK(hipMemcpy(p, d_nbflags.p, nn, hipMemcpyDeviceToHost));
    p += pad8(nn);
    if (nn) HIPCHECK(hipMemcpy(p, d_xN.p, 8 * nn, hipMemcpyDeviceToHost));
    p += 8 * nn;
    if (mode == 1) {
        std::vector<double> g(nn), b(mm);
        if (nn) HIPCHECK(hipMemcpy(g.data(), d_gamma.p, 8 * nn, hipMemcpyDeviceToHost));
        if (mm) HIPCHECK(hipMemcpy(b.data(), d_beta.p, 8 * mm, hipMemcpyDeviceToHost));
        float* f = reinterpret_cast<float*>(p);
        for (size_t i = 0; i < nn; ++i) f[i] = (float)g[i];
        p += pad8(4 * nn);
        f = reinterpret_cast<float*>(p);
        for (size_t i = 0; i < mm; ++i) f[i] = (float)b[i];
        p += pad8(4 * mm);
    } else if (mode == 2) {
        if (mm) HIPCHECK(hipMemcpy(p, d_xB.p, 8 * mm, hipMemcpyDeviceToHost));
        p += 8 * mm;
        if (nn) HIPCHECK(hipMemcpy(p, d_d.p, 8 * nn, hipMemcpyDeviceToHost));
        p += 8 * nn;
        if (nn) HIPCHECK(hipMemcpy(p, d_gamma.p, 8 * nn, hipMemcpyDeviceToHost));
        p += 8 * nn;
        if (mm) HIPCHECK(hipMemcpy(p, d_beta.p, 8 * mm, hipMemcpyDeviceToHost));
        p += 8 * mm;
    }
    return out;
}
void Engine::load_basis(const uint8_t* blob, size_t len) {
    if (shard_world > 1) throw MlpError(-1, "load_basis: not available on a sharded solution");
    if (!blob || len < sizeof(BasisHeader)) throw MlpError(-1, "load_basis: blob too short");
    BasisHeader h;
    std::memcpy(&h, blob, sizeof(h));
    if (std::memcmp(h.magic, kBasisMagic, 8) != 0 || h.version != 1 || h.mode > 2)
        throw MlpError(-1, "load_basis: not a basis blob of this library (magic / version / mode)");
    const size_t mm = (size_t)m_, nn = (size_t)num_vars;
    if (h.m != mm || h.n != nn)
        throw MlpError(-1, "load_basis: the basis belongs to a model with " + std::to_string(h.m) + " kept rows and " +
                               std::to_string(h.n) + " variables, this one has " + std::to_string(mm) + " and " + std::to_string(nn));
    const int mode = (int)h.mode;
    if (len < basis_blob_size(mode, mm, nn)) throw MlpError(-1, "load_basis: blob truncated");
    const uint8_t* p = blob + sizeof(h);
    std::vector<int> bv(mm), nv(nn);
    std::memcpy(bv.data(), p, 4 * mm); p += pad8(4 * mm);
    std::memcpy(nv.data(), p, 4 * nn); p += pad8(4 * nn);
    std::vector<uint8_t> flags(p, p + nn); p += pad8(nn);
    std::vector<double> xN(nn);
    std::memcpy(xN.data(), p, 8 * nn); p += 8 * nn;
    // the two sets must partition the variables
    std::vector<int> loc(N_, INT32_MIN);
    for (size_t r = 0; r < mm; ++r) {
        const int v = bv[r];
        if (v < 0 || v >= N_ || loc[v] != INT32_MIN) throw MlpError(-1, "load_basis: basic_vars is not a set of variables of this model");
        loc[v] = (int)r;
    }
    for (size_t c = 0; c < nn; ++c) {
        const int v = nv[c];
        if (v < 0 || v >= N_ || loc[v] != INT32_MIN) throw MlpError(-1, "load_basis: nb_vars overlaps basic_vars or repeats a variable");
        loc[v] = -1 - (int)c;
    }
    HIPCHECK(hipStreamSynchronize(st));
    h_basic_vars = bv;
    h_nb_vars = nv;
    h_var_loc = loc;
    h_nb_fixed.assign(nn, 0);
    std::vector<double> loB(mm), hiB(mm);
    for (size_t r = 0; r < mm; ++r) { loB[r] = h_lo[bv[r]]; hiB[r] = h_hi[bv[r]]; }
    nnz_nonbasic = 0;
    for (size_t c = 0; c < nn; ++c) {
        // the blob's non-basic values are checked and its at-min / at-max flags RE-DERIVED from them (solver.rs:184-185:
        // exact equality with the bounds); only the fixed marker of fix_var is taken from the blob
        const double x = xN[c], lo = h_lo[nv[c]], hi = h_hi[nv[c]];
        if (!(x >= lo && x <= hi) || !std::isfinite(x))
            throw MlpError(-1, "load_basis: non-basic value " + std::to_string(x) + " of variable " + std::to_string(nv[c]) + " lies outside its bounds");
        const bool fixed = (flags[c] & NB_FIXED) != 0;
        flags[c] = fixed ? (uint8_t)(NB_AT_MIN | NB_AT_MAX | NB_FIXED) : (uint8_t)((x == lo ? NB_AT_MIN : 0) | (x == hi ? NB_AT_MAX : 0));
        h_nb_fixed[c] = fixed ? 1 : 0;
        nnz_nonbasic += (size_t)col_nnz(nv[c]);
    }
    d_basic_vars.upload(h_basic_vars, st); d_nb_vars.upload(h_nb_vars, st); d_var_loc.upload(h_var_loc, st);
    d_loB.upload(loB, st); d_hiB.upload(hiB, st);
    order_valid = false;  // the locality order of the banded sweep belongs to the old non-basic set
    order_force = true;   // a loaded basis is a scrambled one: use the order from the first pivot
    view_dirty = true;
    d_nbflags.upload(flags, st); d_xN.upload(xN, st);
    HIPCHECK(hipStreamSynchronize(st));  // local staging buffers
    sync_view();
    launch_init_nb_rng(hview, geom(), st);
    rebuild_inverse();  // classify singleton / nucleus columns, invert the nucleus from A on the device
    enable_dse = (h.flags & 8u) != 0;
    if (mode == 2) {
        std::vector<double> xB(mm), d(nn), gm(nn), bt(mm);
        std::memcpy(xB.data(), p, 8 * mm); p += 8 * mm;
        std::memcpy(d.data(), p, 8 * nn); p += 8 * nn;
        std::memcpy(gm.data(), p, 8 * nn); p += 8 * nn;
        std::memcpy(bt.data(), p, 8 * mm); p += 8 * mm;
        for (size_t i = 0; i < nn; ++i)
            if (!(gm[i] > 0.0) || !std::isfinite(gm[i]) || !std::isfinite(d[i])) throw MlpError(-1, "load_basis: non-finite reduced cost or non-positive primal edge weight in the blob");
        for (size_t i = 0; i < mm; ++i)
            if (!(bt[i] > 0.0) || !std::isfinite(bt[i]) || !std::isfinite(xB[i])) throw MlpError(-1, "load_basis: non-finite basic value or non-positive dual edge weight in the blob");
        d_xB.upload(xB, st); d_d.upload(d, st); d_gamma.upload(gm, st); d_beta.upload(bt, st);
        HIPCHECK(hipMemcpyAsync(&d_ctl.p->it.obj, &h.obj, sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHECK(hipStreamSynchronize(st));
        primal_feasible = (h.flags & 1u) != 0;
        dual_feasible = (h.flags & 2u) != 0;
        enable_pse = (h.flags & 4u) != 0;
        resume_in_optimize = (h.flags & 16u) != 0;
    } else {
        std::vector<double> gm(nn, 1.0), bt(mm, 1.0);
        if (mode == 1) {
            std::vector<float> f(std::max(nn, mm));  // (the caller's buffer need not be aligned for float reads)
            std::memcpy(f.data(), p, 4 * nn);
            for (size_t i = 0; i < nn; ++i) gm[i] = (double)f[i];
            p += pad8(4 * nn);
            std::memcpy(f.data(), p, 4 * mm);
            for (size_t i = 0; i < mm; ++i) bt[i] = (double)f[i];
            p += pad8(4 * mm);
            for (size_t i = 0; i < nn; ++i)
                if (!(gm[i] > 0.0) || !std::isfinite(gm[i])) throw MlpError(-1, "load_basis: non-positive or non-finite primal edge weight in the blob");
            for (size_t i = 0; i < mm; ++i)
                if (!(bt[i] > 0.0) || !std::isfinite(bt[i])) throw MlpError(-1, "load_basis: non-positive or non-finite dual edge weight in the blob");
        }
        d_gamma.upload(gm, st); d_beta.upload(bt, st);
        HIPCHECK(hipStreamSynchronize(st));
        recalc_basic_vals();                       // x_B = B^-1 (b - N x_N), two refinement steps
        primal_feasible = basic_values_feasible();
        enable_pse = (h.flags & 4u) != 0;
        recalc_obj_coeffs();                       // d and the objective of the real cost vector
        dual_feasible = reduced_costs_feasible();  // re-derived from the recomputed d (the saved bit is not trusted)
        if (!primal_feasible && !dual_feasible)
            throw MlpError(-1, "load_basis: the basis is neither primal nor dual feasible for this model (the artificial-"
                               "objective phase needs a mode-2 checkpoint)");
        resume_in_optimize = primal_feasible && !dual_feasible;
    }
    iters_since_recalc = 0;
    iters_since_polish = 0;
    values_dirty = true;
    budget_exhausted = false;
    beta_stale = false;  // the weights came from the blob (or were reset to 1)
    HIPCHECK(hipStreamSynchronize(st));
}

// ------------------------------------------------------------------ clone (lib.rs:313 / solver.rs:14)
Engine* Engine::clone() {
    flush_lowrank();
    pull_maps();
    std::unique_ptr<Engine> owner(new Engine());  // a throw below (hipMalloc failure, ...) must not leak the half-built clone
    Engine* e = owner.get();
    e->num_vars = num_vars; e->direction = direction;
    e->m_ = m_; e->N_ = N_;
    e->h_obj = h_obj; e->h_lo = h_lo; e->h_hi = h_hi; e->h_rhs = h_rhs;
    e->h_rptr = h_rptr; e->h_rcol = h_rcol; e->h_rval = h_rval;
    e->h_colnnz = h_colnnz; e->h_single_row = h_single_row; e->h_single_val = h_single_val;
    e->max_col_nnz_ = max_col_nnz_; e->max_row_nnz_ = max_row_nnz_; e->amax_ = amax_; e->fpull_on_ = fpull_on_;
    e->sw_balanced = sw_balanced; e->str_kmax = str_kmax; e->sb_kmax = sb_kmax; e->ph_kmax = ph_kmax; e->hyper_mode = hyper_mode; e->hyper_heavy = hyper_heavy; e->hyper_backoff_max = hyper_backoff_max; e->ratio_two = ratio_two; e->ratio_spin_limit = ratio_spin_limit; e->ranks_share_device = false; e->lazy_dse = lazy_dse; e->beta_stale = beta_stale; e->use_order = use_order; e->use_pack = use_pack; e->order_force = order_force; e->lifetime_pivots = lifetime_pivots;
    e->h_basic_vars = h_basic_vars; e->h_nb_vars = h_nb_vars; e->h_var_loc = h_var_loc;
    e->h_kslot_of_pos = h_kslot_of_pos; e->h_srow_of_pos = h_srow_of_pos; e->h_kslot_of_row = h_kslot_of_row;
    e->h_pos_of_srow = h_pos_of_srow; e->h_sdiag_of_pos = h_sdiag_of_pos; e->h_nb_fixed = h_nb_fixed;
    e->h_pos_of_kslot = h_pos_of_kslot; e->h_row_of_kslot = h_row_of_kslot;
    e->enable_pse = enable_pse; e->enable_dse = enable_dse;
    e->primal_feasible = primal_feasible; e->dual_feasible = dual_feasible;
    e->resume_in_optimize = resume_in_optimize;
    e->nnz_nonbasic = nnz_nonbasic;
    e->trace = trace; e->profile = profile;
    e->banded_mode = banded_mode; e->det_mode = det_mode;
    e->final_refresh_pivots = final_refresh_pivots; e->iters_since_recalc = iters_since_recalc;
    e->iters_since_polish = iters_since_polish;
    e->ld_pad = ld_pad; e->lr_force = lr_force; e->force_big_tiles = force_big_tiles;
    hipStream_t s2 = e->st;
    {   // the matrix is copied device to device (the host keeps no CSC)
        const size_t nz = h_rcol.size();
        e->d_cptr.copy_from(d_cptr, (size_t)N_ + 1, s2); e->d_crow.copy_from(d_crow, nz, s2); e->d_cval.copy_from(d_cval, nz, s2);
        e->d_rptr.copy_from(d_rptr, (size_t)m_ + 1, s2); e->d_rcol.copy_from(d_rcol, nz, s2); e->d_rval.copy_from(d_rval, nz, s2);
        e->d_lo.copy_from(d_lo, (size_t)N_, s2); e->d_hi.copy_from(d_hi, (size_t)N_, s2); e->d_obj.copy_from(d_obj, (size_t)N_, s2);
        e->colblk_dirty = true;
        e->banded_dirty = true;
        e->view_dirty = true;
    }
    int mk = m_;
    e->m_ = 0;
    e->alloc_row_buffers(mk);
    e->m_ = mk;
    size_t mm = (size_t)m_, nn = (size_t)num_vars;
    e->d_var_loc.copy_from(d_var_loc, (size_t)N_, s2);
    e->d_basic_vars.copy_from(d_basic_vars, mm, s2);
    e->d_xB.copy_from(d_xB, mm, s2); e->d_loB.copy_from(d_loB, mm, s2); e->d_hiB.copy_from(d_hiB, mm, s2);
    e->d_beta.copy_from(d_beta, mm, s2);
    e->d_nb_vars.copy_from(d_nb_vars, nn, s2); e->d_d.copy_from(d_d, nn, s2); e->d_xN.copy_from(d_xN, nn, s2);
    e->d_gamma.copy_from(d_gamma, nn, s2); e->d_nbflags.copy_from(d_nbflags, nn, s2);
    e->d_alpha_r.ensure(nn, 0, s2); e->d_helper.ensure(nn, 0, s2); e->d_nb_rng.copy_from(d_nb_rng, nn, s2);
    e->ensure_red();
    e->d_ticket.ensure(TK_WORDS, 0, s2);
    HIPCHECK(hipMemsetAsync(e->d_ticket.p, 0, e->d_ticket.cap * sizeof(unsigned), s2));
    e->d_ctl.copy_from(d_ctl, 1, s2);
    // nucleus: same capacity so the block copies 1:1
    e->k_ = 0;
    e->cap_ = 0;
    e->ensure_nucleus_cap(std::min(cap_, std::max(m_, 1)));
    if (e->cap_ != cap_) throw MlpError(-3, "clone: capacity mismatch");
    e->k_ = k_;
    if (k_ > 0)
        HIPCHECK(hipMemcpyAsync(e->d_W.p, d_W.p, sizeof(double) * (size_t)cap_ * ld(), hipMemcpyDeviceToDevice, s2));
    e->push_maps();
    HIPCHECK(hipStreamSynchronize(s2));
    std::memcpy(e->h_ctl, h_ctl, sizeof(Ctl));
    e->values_dirty = true;
    e->fac_mode = fac_mode; e->fac_J_ = fac_J_; e->fac_period_ = fac_period_; e->fac_period_auto_ = fac_period_auto_; e->fac_auto_cap_ = fac_auto_cap_; e->fac_bump_max_ = fac_bump_max_; e->fac_sb_max_ = fac_sb_max_; e->fac_sb_from_ = fac_sb_from_; e->fac_pair_ = fac_pair_; e->fac_skip_ = fac_skip_; e->fac_flow_ = fac_flow_;
    e->cold_start_ = false;  // a clone continues a solve (branch-and-bound: clone + fix_var): warm-start policy, no eager capture of the long graph
    if (fac_on_ && !e->fac_enter())  // (a fresh peel of the same basis: the same operator, no pending terms)
        throw MlpError(-3, "clone: the basis of a solution on the compact factor must peel");
    return owner.release();
}

// ------------------------------------------------------------------ white-box state (tests)
uint64_t Engine::state(const char* what, double* out, uint64_t cap) {
    HIPCHECK(hipStreamSynchronize(st));
    std::string w(what);
    if (w == "dual_edge_sq_norms") ensure_beta();
    std::vector<double> tmp;
    auto from_dev_d = [&](const double* b, size_t n) {
        tmp.resize(n);
        if (n) HIPCHECK(hipMemcpy(tmp.data(), b, n * sizeof(double), hipMemcpyDeviceToHost));
    };
    auto from_dev_i = [&](const DevBuf<int>& b, size_t n) {
        std::vector<int> t(n);
        if (n) HIPCHECK(hipMemcpy(t.data(), b.p, n * sizeof(int), hipMemcpyDeviceToHost));
        tmp.assign(t.begin(), t.end());
    };
    size_t mm = m_, nn = num_vars;
    if (w == "basic_vars") from_dev_i(d_basic_vars, mm);
    else if (w == "basic_var_vals") from_dev_d(d_xB.p, mm);
    else if (w == "basic_var_mins") from_dev_d(d_loB.p, mm);
    else if (w == "basic_var_maxs") from_dev_d(d_hiB.p, mm);
    else if (w == "dual_edge_sq_norms") from_dev_d(d_beta.p, mm);
    else if (w == "nb_vars") from_dev_i(d_nb_vars, nn);
    else if (w == "nb_var_obj_coeffs") from_dev_d(d_d.p, nn);
    else if (w == "nb_var_vals") from_dev_d(d_xN.p, nn);
    else if (w == "primal_edge_sq_norms") from_dev_d(d_gamma.p, nn);
    else if (w == "var_loc") from_dev_i(d_var_loc, (size_t)N_);
    else if (w == "col_coeffs") from_dev_d(d_work.p, mm);
    else if (w == "row_coeffs") from_dev_d(d_alpha_r.p, nn);
    else if (w == "tau") from_dev_d(d_work.p + mm, mm);
    else if (w == "inv_basis_row_coeffs" || w == "v") {  // (rho, v) interleaved by row (solver.rs:56, 1114)
        std::vector<double> both(2 * mm);
        if (mm) HIPCHECK(hipMemcpy(both.data(), d_work.p + 2 * mm, 2 * mm * sizeof(double), hipMemcpyDeviceToHost));
        tmp.resize(mm);
        const size_t off = (w == "v") ? 1 : 0;
        for (size_t i = 0; i < mm; ++i) tmp[i] = both[2 * i + off];
    } else if (w == "sq_norms_update_helper") from_dev_d(d_helper.p, nn);
    else if (w == "nb_flags") {
        std::vector<uint8_t> t(nn);
        if (nn) HIPCHECK(hipMemcpy(t.data(), d_nbflags.p, nn, hipMemcpyDeviceToHost));
        tmp.assign(t.begin(), t.end());
    } else if (w == "cur_obj_val") tmp = {cur_obj_val()};
    else if (w == "orig_obj_coeffs") tmp = h_obj;
    else if (w == "orig_var_mins") tmp = h_lo;
    else if (w == "orig_var_maxs") tmp = h_hi;
    else if (w == "orig_rhs") tmp = h_rhs;
    else if (w == "flags") tmp = {(double)primal_feasible, (double)dual_feasible, (double)enable_pse, (double)enable_dse};
    else if (w == "small_basis_launches") {  // iterations that ran BTRAN + pass + v tail + touch as one launch (k_small_basis)
        pull_ctl();
        tmp = {(double)h_ctl->sb_count};
    } else if (w == "dual_list_tests") {  // dual Harris tests that ran over the listed non-zeros of alpha_r (kernels.hip ratio_dual_list)
        int n_ = 0, p_ = 0;  // [tests over a list, times a dense row paused the listing]
        HIPCHECK(hipMemcpy(&n_, &d_ctl.p->ar_used, sizeof(int), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(&p_, &d_ctl.p->ar_pauses, sizeof(int), hipMemcpyDeviceToHost));
        tmp = {(double)n_, (double)p_};
    } else if (w == "fpull") {  // pulled F product: [in use now, builds of the packed copy, pivot count at the last build]
        tmp = {(double)(hview.fpk_on ? 1 : 0), (double)fpk_builds_, (double)fpk_built_at_, (double)(fpull_supported(hview, geom()) ? 1 : 0)};
    } else if (w == "golive_checks") {  // fingerprint comparisons passed at the go-live point of the deferred sharding
        tmp = {(double)golive_checks_};
    } else if (w == "shard_live") {
        tmp = {(double)(shard_is_live() ? 1 : 0)};
    } else if (w == "primal_head_launches") {
        pull_ctl();
        tmp = {(double)h_ctl->ph_count};
    }
    else if (w == "hyper_profile") {  // microseconds per stage of the hypersparse iteration, accumulated since try_new
        pull_ctl();
        for (int i = 0; i < 13; ++i) tmp.push_back((double)h_ctl->hy_prof[i] * 0.01);
        tmp.push_back((double)h_ctl->hy_prof[13]);  // shader-clock cycles of the launches (with slot 12: the clock the kernel ran at)
        tmp.push_back((double)h_ctl->hy_prof[14] * 0.01);  // prologues
        tmp.push_back(0.0);
        for (int i = 16; i < 24; ++i) tmp.push_back((double)h_ctl->hy_prof[i] * 0.01);  // sub-stage marks (experiments)
    }
    else if (w == "kernel_timeline") {  // MLP_KPROF=1: wall-clock marks of the last iteration, microseconds relative to the earliest mark
        pull_ctl();
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < 24; ++i) if (h_ctl->hy_prof[i] && h_ctl->hy_prof[i] < t0) t0 = h_ctl->hy_prof[i];
        for (int i = 0; i < 24; ++i) tmp.push_back(h_ctl->hy_prof[i] ? (double)(h_ctl->hy_prof[i] - t0) * 0.01 : -1.0);
    }
    else if (w == "factor_plan") {  // compact factor: fac_meta[0..8), then the segments of the walk (kind, first level, last level) and their positions
        if (fac_on_ && d_fac_meta.p && d_fac_segs.p) {
            int meta[8] = {0};
            HIPCHECK(hipStreamSynchronize(st));
            HIPCHECK(hipMemcpy(meta, d_fac_meta.p, sizeof(meta), hipMemcpyDeviceToHost));
            const int nseg = std::max(0, std::min(meta[7], FAC_MAX_LEVELS));
            std::vector<int> sg((size_t)3 * nseg), lp((size_t)meta[0] + 1);
            if (nseg) HIPCHECK(hipMemcpy(sg.data(), d_fac_segs.p, sizeof(int) * sg.size(), hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(lp.data(), d_fac_lptr.p, sizeof(int) * lp.size(), hipMemcpyDeviceToHost));
            for (int i = 0; i < 8; ++i) tmp.push_back((double)meta[i]);
            for (int q = 0; q < nseg; ++q) {
                tmp.push_back((double)sg[3 * q]); tmp.push_back((double)sg[3 * q + 1]); tmp.push_back((double)sg[3 * q + 2]);
                tmp.push_back((double)(lp[sg[3 * q + 2] + 1] - lp[sg[3 * q + 1]]));
            }
        }
    }
    else if (w == "factor_sb") {  // sparse factor of the bump: in use now, factorisations, fallbacks to the dense inverse, rounds of the last one
        tmp.push_back(fac_on_ && fac_sb_on_ ? 1.0 : 0.0); tmp.push_back((double)stats.fac_sb_factors);
        tmp.push_back((double)stats.fac_sb_fallbacks); tmp.push_back((double)stats.fac_sb_rounds); tmp.push_back((double)stats.fac_sb_tail);
        tmp.push_back((double)stats.fac_sb_skipped); tmp.push_back((double)fac_sb_fail_b_);  // refactorisations that skipped the sparse attempt, size of the bump that failed last
    }
    else if (w == "hyper_bail_reasons") {
        for (int i = 0; i < 10; ++i) tmp.push_back((double)stats.hyper_bail_reason[i]);
    } else if (w == "w_checksum") {  // (tests of the sharded path) checksum of the nucleus inverse after folding the pending terms: [low 32 bits, high 32 bits, k]
        flush_lowrank();
        sync_view();
        DevBuf<unsigned long long> cs;
        cs.ensure(1, 0, st);
        HIPCHECK(hipMemsetAsync(cs.p, 0, sizeof(unsigned long long), st));
        if (k_ > 0) launch_checksum_w(hview, cs.p, st);
        unsigned long long h = 0;
        HIPCHECK(hipMemcpyAsync(&h, cs.p, sizeof(h), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        tmp = {(double)(h & 0xffffffffull), (double)(h >> 32), (double)k_};
    } else if (w == "reinvert_scale") tmp = {last_reinvert_scale};  // max |entry| of the fresh nucleus inverse of the last reinvert()
    else if (w == "host_basic_vars") tmp.assign(h_basic_vars.begin(), h_basic_vars.end());
    else if (w == "host_nb_vars") tmp.assign(h_nb_vars.begin(), h_nb_vars.end());
    else return (uint64_t)-1;
    if (out) for (size_t i = 0; i < tmp.size() && i < cap; ++i) out[i] = tmp[i];
    return tmp.size();
}

}  // namespace mlp
