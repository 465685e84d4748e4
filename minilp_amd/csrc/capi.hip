// capi.hip — extern "C" boundary (include/minilp_hip.h).  No C++ exception crosses the ABI.
#include "../../include/minilp_hip.h"

#include <cmath>
#include <cstring>
#include <string>

#include "engine.h"

using namespace mlp;

struct mlp_problem {
    ProblemData pd;
};
struct mlp_solution {
    Engine* eng = nullptr;
    ~mlp_solution() { delete eng; }
};
struct mlp_mps {
    MpsData d;
};

static thread_local std::string g_err;

template <class F>
static int guarded(F f) {
    try {
        f();
        return MLP_OK;
    } catch (LpFail& e) {
        return e.code;
    } catch (MlpError& e) {
        g_err = e.what();
        return e.code;
    } catch (std::exception& e) {
        g_err = e.what();
        return MLP_EINVAL;
    }
}
static void refuse_if_sharded(mlp_solution* s) {
    if (s->eng->sharded()) throw MlpError(MLP_EINVAL, "not available on a sharded solution");
}
static void require(mlp_solution** s) {
    if (!s || !*s) throw MlpError(MLP_EINVAL, "NULL solution (consumed by a failed mutator?)");
}
static int consume_on_error(mlp_solution** s, int st) {  // lib.rs:359, 385
    if (st != 0 && s && *s) {
        delete *s;
        *s = nullptr;
    }
    return st;
}

extern "C" {

const char* mlp_last_error(void) { return g_err.c_str(); }
int mlp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mlp_set_device(int device) {
    return guarded([&] { HIPCHECK(hipSetDevice(device)); });
}

mlp_problem* mlp_problem_new(int direction) {
    mlp_problem* p = new mlp_problem();
    p->pd.direction = direction == MLP_MAXIMIZE ? 1 : 0;
    return p;
}
mlp_problem* mlp_problem_clone(const mlp_problem* p) { return new mlp_problem(*p); }
void mlp_problem_free(mlp_problem* p) { delete p; }
uint32_t mlp_problem_add_var(mlp_problem* p, double c, double mn, double mx) { return (uint32_t)p->pd.add_var(c, mn, mx); }
uint32_t mlp_problem_num_vars(const mlp_problem* p) { return (uint32_t)p->pd.obj.size(); }
int mlp_problem_add_constraint(mlp_problem* p, const uint32_t* vars, const double* coeffs, uint64_t k, int op, double rhs) {
    return guarded([&] { p->pd.add_constraint(vars, coeffs, k, op, rhs); });
}
int mlp_problem_add_vars(mlp_problem* p, uint64_t n, const double* obj, const double* mins, const double* maxs) {
    return guarded([&] {
        for (uint64_t i = 0; i < n; ++i) p->pd.add_var(obj[i], mins[i], maxs[i]);
    });
}
int mlp_problem_add_constraints_csr(mlp_problem* p, uint64_t m, const uint64_t* indptr, const uint32_t* vars,
                                    const double* coeffs, const int32_t* ops, const double* rhs) {
    return guarded([&] {
        for (uint64_t i = 0; i < m; ++i)
            p->pd.add_constraint(vars + indptr[i], coeffs + indptr[i], indptr[i + 1] - indptr[i], ops[i], rhs[i]);
    });
}
uint64_t mlp_problem_num_constraints(const mlp_problem* p) { return p->pd.cons.size(); }
int mlp_problem_var(const mlp_problem* p, uint32_t var, double* obj, double* mn, double* mx) {
    return guarded([&] {
        if (var >= p->pd.obj.size()) throw MlpError(MLP_EINVAL, "variable out of range");
        *obj = p->pd.direction == 1 ? -p->pd.obj[var] : p->pd.obj[var];
        *mn = p->pd.lo[var];
        *mx = p->pd.hi[var];
    });
}
uint64_t mlp_problem_constraint(const mlp_problem* p, uint64_t ci, uint32_t* vars, double* coeffs, uint64_t cap, int* op, double* rhs) {
    if (ci >= p->pd.cons.size()) return 0;
    const Constraint& c = p->pd.cons[ci];
    if (op) *op = c.op;
    if (rhs) *rhs = c.rhs;
    for (size_t i = 0; i < c.idx.size() && i < cap; ++i) {
        vars[i] = (uint32_t)c.idx[i];
        coeffs[i] = c.val[i];
    }
    return c.idx.size();
}
int mlp_problem_solve_ex(const mlp_problem* p, mlp_solution** out, int64_t budget, uint32_t flags) {
    *out = nullptr;
    mlp_solution* s = new mlp_solution();
    int st = guarded([&] {
        s->eng = new Engine();
        s->eng->pivot_budget = budget;
        s->eng->trace = flags & 1u;
        s->eng->profile = flags & 2u;
        s->eng->sample_every = (flags & 4u) ? 1 : 0;
        s->eng->try_new(p->pd);      // lib.rs:292-297
        s->eng->initial_solve();     // lib.rs:298
    });
    if (st != 0) delete s;
    else *out = s;
    return st;
}
int mlp_problem_solve(const mlp_problem* p, mlp_solution** out) { return mlp_problem_solve_ex(p, out, -1, 0); }
int mlp_problem_solve_from_basis(const mlp_problem* p, const void* blob, uint64_t len, mlp_solution** out, int64_t budget,
                                 uint32_t flags) {
    *out = nullptr;
    mlp_solution* s = new mlp_solution();
    int st = guarded([&] {
        s->eng = new Engine();
        s->eng->pivot_budget = budget;
        s->eng->trace = flags & 1u;
        s->eng->profile = flags & 2u;
        s->eng->sample_every = (flags & 4u) ? 1 : 0;
        s->eng->try_new(p->pd);
        s->eng->load_basis(static_cast<const uint8_t*>(blob), (size_t)len);
        s->eng->initial_solve();
    });
    if (st != 0) delete s;
    else *out = s;
    return st;
}
uint64_t mlp_solution_save_basis(const mlp_solution* s, int mode, void* buf, uint64_t cap) {
    uint64_t need = 0;
    int st = guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution");
        std::vector<uint8_t> b = s->eng->save_basis(mode);
        need = b.size();
        if (buf && cap >= need) std::memcpy(buf, b.data(), b.size());
    });
    return st == 0 ? need : 0;
}
int mlp_solution_set_sampling(mlp_solution* s, int every_iteration) {
    return guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution");
        if (every_iteration < 0) {  // switch the HIP-event sampling off altogether (long runs after a measurement pass)
            s->eng->profile = false;
            s->eng->sample_every = 0;
        } else {
            s->eng->profile = true;  // (also switches the sampling back on after a `< 0` call)
            s->eng->sample_every = every_iteration ? 1 : 0;
        }
    });
}
int mlp_solution_continue(mlp_solution* s, int64_t budget) {
    return guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution");
        s->eng->pivot_budget = budget;
        s->eng->budget_exhausted = false;
        s->eng->initial_solve();
    });
}
static void copy_info(const Engine::StepInfo& si, mlp_iter_info* out) {
    if (!out) return;
    out->status = si.status; out->phase = si.phase; out->next_stage = si.next_stage; out->reserved = 0;
    out->col = si.col; out->row = si.row; out->entering_var = si.entering_var; out->leaving_var = si.leaving_var;
    out->pivot_coeff = si.pivot_coeff; out->step = si.step; out->objective = si.objective;
    out->nucleus_size = si.nucleus_size;
}
int mlp_engine_open(mlp_solution* s, mlp_iter_info* out) {
    int status = MLP_EINVAL;
    int rc = guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution");
        Engine::StepInfo si{};
        status = s->eng->step_open(&si);
        copy_info(si, out);
    });
    return rc != 0 ? (rc > 0 ? MLP_EINVAL : rc) : status;
}
int mlp_engine_stage(mlp_solution* s, int stage, mlp_iter_info* out) {
    int status = MLP_EINVAL;
    int rc = guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution");
        Engine::StepInfo si{};
        status = s->eng->step_stage(stage, &si);
        copy_info(si, out);
    });
    return rc != 0 ? (rc > 0 ? MLP_EINVAL : rc) : status;
}
int mlp_solution_budget_exhausted(const mlp_solution* s) { return (s && s->eng->budget_exhausted) ? 1 : 0; }
int mlp_solution_reinvert(mlp_solution* s, double* max_diff) {
    return guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution");
        double d = s->eng->reinvert(true);
        if (max_diff) *max_diff = d;
    });
}

int mlp_solution_recompute_basic_values(mlp_solution* s) {
    return guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution");
        // x_B, then the objective of the recomputed point (and the reduced costs with it: solver.rs:1199-1231), so that
        // mlp_solution_objective and the values agree afterwards
        // — except inside the artificial-objective feasibility phase of a budget-paused solve (d = +-1 / 0 there: initial_solve
        // itself never recomputes d on a budget resume); the basic values are recomputed, d and the objective are left alone
        s->eng->recalc_basic_vals();
        if (!s->eng->in_artificial_phase()) s->eng->refresh_objective();
    });
}
uint32_t mlp_abi_version(void) { return MLP_ABI_VERSION; }
uint64_t mlp_stats_size(void) { return (uint64_t)sizeof(mlp_stats); }

int mlp_solution_enable_sharding(mlp_solution* s, int rank, int world, const char* shm_name) {
    return guarded([&] {
        if (!s || !shm_name) throw MlpError(MLP_EINVAL, "NULL solution / rendezvous name");
        s->eng->enable_sharding(rank, world, shm_name);
    });
}

int mlp_solution_enable_sharding_ex(mlp_solution* s, int rank, int world, const char* shm_name, const char* transport, const void* rccl_id) {
    return guarded([&] {
        if (!s || !shm_name) throw MlpError(MLP_EINVAL, "NULL solution / rendezvous name");
        s->eng->enable_sharding(rank, world, shm_name, transport, rccl_id);
    });
}
int mlp_rccl_unique_id(void* out128) {
    return guarded([&] {
        if (!out128) throw MlpError(MLP_EINVAL, "NULL buffer");
        Engine::rccl_unique_id(out128);
    });
}
const char* mlp_solution_transport(const mlp_solution* s) { return s ? s->eng->transport.c_str() : "none"; }

mlp_solution* mlp_solution_clone(const mlp_solution* s) {
    mlp_solution* c = new mlp_solution();
    int st = guarded([&] { c->eng = s->eng->clone(); });
    if (st != 0) {
        delete c;
        return nullptr;
    }
    return c;
}
void mlp_solution_free(mlp_solution* s) { delete s; }
double mlp_solution_objective(const mlp_solution* s) {  // lib.rs:334-339
    if (!s) return std::nan("");
    double v = 0.0;
    if (guarded([&] { v = s->eng->cur_obj_val(); }) != MLP_OK) return std::nan("");  // HIP failure: never a plausible 0
    return s->eng->direction == 1 ? -v : v;
}
uint32_t mlp_solution_num_vars(const mlp_solution* s) { return s ? (uint32_t)s->eng->num_vars : 0; }
int mlp_solution_var_value(const mlp_solution* s, uint32_t var, double* out) {
    return guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution (consumed by a failed mutator?)");
        if ((int)var >= s->eng->num_vars) throw MlpError(MLP_EINVAL, "variable out of range (lib.rs:345)");
        *out = s->eng->get_value((int)var);
    });
}
int mlp_solution_values(const mlp_solution* s, double* out, uint32_t n) {
    return guarded([&] {
        if (!s) throw MlpError(MLP_EINVAL, "NULL solution (consumed by a failed mutator?)");
        if ((int)n > s->eng->num_vars) throw MlpError(MLP_EINVAL, "too many values requested");
        s->eng->get_values(out, (int)n);
    });
}
int mlp_solution_add_constraint(mlp_solution** s, const uint32_t* vars, const double* coeffs, uint64_t k, int op, double rhs) {
    return consume_on_error(s, guarded([&] {
        require(s);
        ProblemData tmp;
        tmp.obj.resize((*s)->eng->num_vars);  // lib.rs:376: dimension = num_vars
        tmp.add_constraint(vars, coeffs, k, op, rhs);
        refuse_if_sharded(*s);
        (*s)->eng->pivot_budget = -1;
        (*s)->eng->add_constraint(tmp.cons[0]);
    }));
}
int mlp_solution_fix_var(mlp_solution** s, uint32_t var, double val) {
    return consume_on_error(s, guarded([&] {
        require(s);
        if ((int)var >= (*s)->eng->num_vars) throw MlpError(MLP_EINVAL, "variable out of range (lib.rs:391)");
        refuse_if_sharded(*s);
        (*s)->eng->pivot_budget = -1;
        (*s)->eng->fix_var((int)var, val);
    }));
}
int mlp_solution_unfix_var(mlp_solution** s, uint32_t var, int* was_fixed) {
    return consume_on_error(s, guarded([&] {
        require(s);
        if ((int)var >= (*s)->eng->num_vars) throw MlpError(MLP_EINVAL, "variable out of range (lib.rs:400)");
        refuse_if_sharded(*s);
        (*s)->eng->pivot_budget = -1;
        *was_fixed = (*s)->eng->unfix_var((int)var) ? 1 : 0;
    }));
}
int mlp_solution_add_gomory_cut(mlp_solution** s, uint32_t var) {
    return consume_on_error(s, guarded([&] {
        require(s);
        if ((int)var >= (*s)->eng->num_vars) throw MlpError(MLP_EINVAL, "variable out of range (lib.rs:420)");
        refuse_if_sharded(*s);
        (*s)->eng->pivot_budget = -1;
        (*s)->eng->add_gomory_cut((int)var);
    }));
}

void mlp_solution_stats(const mlp_solution* s, mlp_stats* o) {
    if (!o) return;
    if (!s) {
        std::memset(o, 0, sizeof(*o));
        return;
    }
    Engine* e = s->eng;
    const Stats& t = e->stats;
    std::memset(o, 0, sizeof(*o));
    o->iterations = t.iterations; o->basis_changes = t.basis_changes; o->bound_flips = t.bound_flips;
    o->primal_iters = t.primal_iters; o->dual_iters = t.dual_iters; o->reinversions = t.reinversions;
    o->num_constraints = e->m(); o->num_total_vars = e->total_vars();
    o->nucleus_size = e->nucleus(); o->nucleus_capacity = e->nucleus_cap(); o->nnz = e->nnz();
    o->fused_bytes = t.fused_bytes; o->fused_ms = t.fused_ms; o->sweep_bytes = t.sweep_bytes; o->sweep_ms = t.sweep_ms;
    o->fused_launches = t.fused_launches; o->sweep_launches = t.sweep_launches;
    o->update_ms = t.update_ms; o->update_launches = t.update_launches; o->final_refreshes = t.final_refreshes;
    o->banded_sweep = e->banded_active() ? 1 : 0;
    o->solve_wall_s = t.solve_wall_s;
    o->max_pivot_err = t.max_pivot_err;
    o->ftran_bytes = t.ftran_bytes; o->ftran_ms = t.ftran_ms; o->ftran_launches = t.ftran_launches;
    o->iter_ms = t.iter_ms; o->iter_samples = t.iter_samples;
    o->beta_rebuilds = t.beta_rebuilds;
    o->str_ms = t.str_ms; o->str_launches = t.str_launches;
    o->ratio_stalls = t.ratio_stalls; o->hyper_iters = t.hyper_iters; o->hyper_bails = t.hyper_bails;
    o->dense_ftran_bytes = t.dense_ftran_bytes; o->dense_ftran_ms = t.dense_ftran_ms; o->dense_ftran_launches = t.dense_ftran_launches;
    o->fold_bytes = t.fold_bytes; o->fold_ms = t.fold_ms; o->fold_launches = t.fold_launches;
    for (int i = 0; i < 5; ++i) o->kase[i] = t.kase[i];
    o->reinversion_fallbacks = t.reinversion_fallbacks;
    o->factor_active = e->factor_active() ? 1 : 0; o->factor_refactors = t.fac_refactors; o->factor_levels = t.fac_levels;
    o->factor_switches = t.fac_switches; o->factor_bump = t.fac_bump; o->factor_bump_max = t.fac_bump_max;
}
void mlp_solution_reset_stats(mlp_solution* s) {
    guarded([&] { s->eng->resolve_events(); });
    s->eng->stats = Stats();
    s->eng->restart_sampling();  // profile mode: the next batch is a sampled one, so a short timed region still has samples
}
uint64_t mlp_solution_trace_len(const mlp_solution* s) { return s->eng->trace_log.size(); }
void mlp_solution_trace_get(const mlp_solution* s, uint64_t i, int32_t* phase, int64_t* col, int64_t* row,
                            int64_t* entering_var, int64_t* leaving_var, double* pivot_coeff, double* obj_after) {
    const PivotRecord& r = s->eng->trace_log[i];
    *phase = r.phase; *col = r.col; *row = r.row; *entering_var = r.entering_var; *leaving_var = r.leaving_var;
    *pivot_coeff = r.pivot_coeff; *obj_after = r.obj_after;
}
uint64_t mlp_solution_state(const mlp_solution* s, const char* what, double* out, uint64_t cap) {
    uint64_t n = (uint64_t)-1;
    guarded([&] { n = s->eng->state(what, out, cap); });
    return n;
}

int mlp_mps_parse(const char* text, uint64_t len, int direction, mlp_mps** out) {
    *out = nullptr;
    mlp_mps* f = new mlp_mps();
    int st = guarded([&] { f->d = parse_mps(std::string(text, len), direction == MLP_MAXIMIZE ? 1 : 0); });
    if (st != 0) delete f;
    else *out = f;
    return st;
}
void mlp_mps_free(mlp_mps* f) { delete f; }
const char* mlp_mps_name(const mlp_mps* f) { return f->d.name.c_str(); }
uint32_t mlp_mps_num_vars(const mlp_mps* f) { return (uint32_t)f->d.var_names.size(); }
const char* mlp_mps_var_name(const mlp_mps* f, uint32_t i) { return f->d.var_names[i].c_str(); }
int64_t mlp_mps_var_index(const mlp_mps* f, const char* name) {
    for (size_t i = 0; i < f->d.var_names.size(); ++i)
        if (f->d.var_names[i] == name) return (int64_t)i;
    return -1;
}
mlp_problem* mlp_mps_problem(const mlp_mps* f) {
    mlp_problem* p = new mlp_problem();
    p->pd = f->d.problem;
    return p;
}

}  // extern "C"
