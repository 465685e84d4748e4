//! Raw FFI declarations of `include/minilp_hip.h` (one-to-one; see that header for the reference lines each
//! entry replaces).  NOT COMPILED in the build environment of this repository (no Rust toolchain there).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int};

#[repr(C)] pub struct mlp_problem { _p: [u8; 0] }
#[repr(C)] pub struct mlp_solution { _p: [u8; 0] }
#[repr(C)] pub struct mlp_mps { _p: [u8; 0] }

pub const MLP_MINIMIZE: c_int = 0;
pub const MLP_MAXIMIZE: c_int = 1;
pub const MLP_EQ: c_int = 0;
pub const MLP_LE: c_int = 1;
pub const MLP_GE: c_int = 2;
pub const MLP_OK: c_int = 0;
pub const MLP_INFEASIBLE: c_int = 1;
pub const MLP_UNBOUNDED: c_int = 2;

pub const MLP_STAGE_FTRAN: c_int = 0;
pub const MLP_STAGE_RATIO: c_int = 1;
pub const MLP_STAGE_BTRAN: c_int = 2;
pub const MLP_STAGE_BASIS: c_int = 3;
pub const MLP_STAGE_ROW: c_int = 4;
pub const MLP_STAGE_APPLY: c_int = 5;
pub const MLP_ITER_PIVOT: c_int = 0;
pub const MLP_ITER_FLIP: c_int = 1;
pub const MLP_ITER_OPTIMAL: c_int = 2;
pub const MLP_ITER_UNBOUNDED: c_int = 3;
pub const MLP_ITER_FEASIBLE: c_int = 4;
pub const MLP_ITER_INFEASIBLE: c_int = 5;
pub const MLP_ITER_SINGULAR: c_int = 6;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct mlp_iter_info {
    pub status: i32,
    pub phase: i32,
    pub next_stage: i32,
    pub reserved: i32,
    pub col: i64,
    pub row: i64,
    pub entering_var: i64,
    pub leaving_var: i64,
    pub pivot_coeff: c_double,
    pub step: c_double,
    pub objective: c_double,
    pub nucleus_size: u64,
}

extern "C" {
    pub fn mlp_last_error() -> *const c_char;
    pub fn mlp_device_count() -> c_int;
    pub fn mlp_set_device(device: c_int) -> c_int;

    pub fn mlp_problem_new(direction: c_int) -> *mut mlp_problem;
    pub fn mlp_problem_clone(p: *const mlp_problem) -> *mut mlp_problem;
    pub fn mlp_problem_free(p: *mut mlp_problem);
    pub fn mlp_problem_add_var(p: *mut mlp_problem, obj: c_double, min: c_double, max: c_double) -> u32;
    pub fn mlp_problem_add_constraint(p: *mut mlp_problem, vars: *const u32, coeffs: *const c_double, k: u64,
                                      cmp_op: c_int, rhs: c_double) -> c_int;
    pub fn mlp_problem_add_vars(p: *mut mlp_problem, n: u64, obj: *const c_double, mins: *const c_double,
                                maxs: *const c_double) -> c_int;
    pub fn mlp_problem_add_constraints_csr(p: *mut mlp_problem, m: u64, indptr: *const u64, vars: *const u32,
                                           coeffs: *const c_double, cmp_ops: *const i32, rhs: *const c_double) -> c_int;
    pub fn mlp_problem_num_constraints(p: *const mlp_problem) -> u64;
    pub fn mlp_problem_var(p: *const mlp_problem, var: u32, obj_coeff: *mut c_double, min: *mut c_double, max: *mut c_double) -> c_int;
    pub fn mlp_problem_constraint(p: *const mlp_problem, c: u64, vars: *mut u32, coeffs: *mut c_double, cap: u64,
                                  cmp_op: *mut c_int, rhs: *mut c_double) -> u64;
    pub fn mlp_problem_solve(p: *const mlp_problem, out: *mut *mut mlp_solution) -> c_int;
    pub fn mlp_problem_solve_ex(p: *const mlp_problem, out: *mut *mut mlp_solution, pivot_budget: i64, flags: u32) -> c_int;

    pub fn mlp_solution_clone(s: *const mlp_solution) -> *mut mlp_solution;
    pub fn mlp_solution_free(s: *mut mlp_solution);
    pub fn mlp_solution_objective(s: *const mlp_solution) -> c_double;
    pub fn mlp_solution_num_vars(s: *const mlp_solution) -> u32;
    pub fn mlp_solution_var_value(s: *const mlp_solution, var: u32, out: *mut c_double) -> c_int;
    pub fn mlp_solution_values(s: *const mlp_solution, out: *mut c_double, n: u32) -> c_int;
    pub fn mlp_solution_add_constraint(s: *mut *mut mlp_solution, vars: *const u32, coeffs: *const c_double, k: u64,
                                       cmp_op: c_int, rhs: c_double) -> c_int;
    pub fn mlp_solution_fix_var(s: *mut *mut mlp_solution, var: u32, val: c_double) -> c_int;
    pub fn mlp_solution_unfix_var(s: *mut *mut mlp_solution, var: u32, was_fixed: *mut c_int) -> c_int;
    pub fn mlp_solution_add_gomory_cut(s: *mut *mut mlp_solution, var: u32) -> c_int;
    pub fn mlp_solution_continue(s: *mut mlp_solution, pivot_budget: i64) -> c_int;
    pub fn mlp_solution_budget_exhausted(s: *const mlp_solution) -> c_int;
    pub fn mlp_solution_reinvert(s: *mut mlp_solution, max_diff: *mut c_double) -> c_int;
    pub fn mlp_solution_recompute_basic_values(s: *mut mlp_solution) -> c_int;
    pub fn mlp_solution_save_basis(s: *const mlp_solution, mode: c_int, buf: *mut std::os::raw::c_void, cap: u64) -> u64;
    pub fn mlp_problem_solve_from_basis(p: *const mlp_problem, blob: *const std::os::raw::c_void, len: u64,
                                        out: *mut *mut mlp_solution, pivot_budget: i64, flags: u32) -> c_int;
    pub fn mlp_solution_enable_sharding(s: *mut mlp_solution, rank: c_int, world: c_int, shm_name: *const c_char) -> c_int;
    pub fn mlp_solution_enable_sharding_ex(s: *mut mlp_solution, rank: c_int, world: c_int, shm_name: *const c_char,
                                           transport: *const c_char, rccl_id: *const std::os::raw::c_void) -> c_int;
    pub fn mlp_rccl_unique_id(out128: *mut std::os::raw::c_void) -> c_int;
    pub fn mlp_solution_transport(s: *const mlp_solution) -> *const c_char;
    pub fn mlp_abi_version() -> u32;
    pub fn mlp_stats_size() -> u64;

    pub fn mlp_engine_open(s: *mut mlp_solution, out: *mut mlp_iter_info) -> c_int;
    pub fn mlp_engine_stage(s: *mut mlp_solution, stage: c_int, out: *mut mlp_iter_info) -> c_int;

    pub fn mlp_mps_parse(text: *const c_char, len: u64, direction: c_int, out: *mut *mut mlp_mps) -> c_int;
    pub fn mlp_mps_free(f: *mut mlp_mps);
    pub fn mlp_mps_name(f: *const mlp_mps) -> *const c_char;
    pub fn mlp_mps_num_vars(f: *const mlp_mps) -> u32;
    pub fn mlp_mps_var_name(f: *const mlp_mps, i: u32) -> *const c_char;
    pub fn mlp_mps_var_index(f: *const mlp_mps, name: *const c_char) -> i64;
    pub fn mlp_mps_problem(f: *const mlp_mps) -> *mut mlp_problem;
}
