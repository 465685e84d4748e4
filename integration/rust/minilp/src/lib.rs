//! `minilp`'s public API (ztlpn/minilp 0.2.2, `src/lib.rs:61-464`) over `libminilp_hip.so`.
//!
//! Every public item of the reference crate is here with the same name, signature and error behaviour; the
//! solver behind `Problem::solve` and the `Solution` mutators is the device-resident simplex engine.  A `Problem`
//! is plain host data exactly as in the reference (`lib.rs:193-200`) and is handed to the library at `solve()`;
//! a `Solution` owns an opaque `mlp_solution` (the state of `solver.rs:14-58` in HBM).
//!
//! Differences by construction: a process without a visible GPU gets a panic from `solve()` (there is no CPU
//! fallback); `Solution: Clone` deep-copies device state.
//!
//! NOT COMPILED in this repository (no `cargo` in the build image).  The same C ABI is exercised by the Python
//! `ctypes` mirror (`minilp_amd/api.py`) that the GPU parity tests go through, call for call.
#![deny(missing_docs)]

use minilp_hip_sys as sys;
use std::cell::RefCell;
use std::ffi::CStr;
use std::ptr;

mod mps;
pub use mps::MpsFile;

/// An enum indicating whether to minimize or maximize objective function.  (`lib.rs:62-69`)
#[derive(Clone, Copy, Debug)]
pub enum OptimizationDirection {
    /// Minimize the objective function.
    Minimize,
    /// Maximize the objective function.
    Maximize,
}

/// A reference to a variable in a linear programming problem.  (`lib.rs:71-83`)
#[derive(Clone, Copy, Debug, PartialEq, Eq, PartialOrd, Ord, Hash)]
pub struct Variable(pub(crate) usize);

impl Variable {
    /// Sequence number of the variable.
    pub fn idx(&self) -> usize {
        self.0
    }
}

/// A sum of variables multiplied by constant coefficients.  (`lib.rs:85-114`)
#[derive(Clone, Debug)]
pub struct LinearExpr {
    vars: Vec<usize>,
    coeffs: Vec<f64>,
}

impl LinearExpr {
    /// Creates an empty linear expression.
    pub fn empty() -> Self {
        Self { vars: vec![], coeffs: vec![] }
    }

    /// Add a single term to the linear expression.  Variables can be added only once (checked at
    /// `add_constraint`, like the reference's `CsVec::new` panic).
    pub fn add(&mut self, var: Variable, coeff: f64) {
        self.vars.push(var.0);
        self.coeffs.push(coeff);
    }
}

/// A single `variable * constant` term in a linear expression.  (`lib.rs:116-129`)
#[derive(Clone, Copy, Debug)]
pub struct LinearTerm(Variable, f64);

impl From<(Variable, f64)> for LinearTerm {
    fn from(term: (Variable, f64)) -> Self {
        LinearTerm(term.0, term.1)
    }
}

impl<'a> From<&'a (Variable, f64)> for LinearTerm {
    fn from(term: &'a (Variable, f64)) -> Self {
        LinearTerm(term.0, term.1)
    }
}

impl<I: IntoIterator<Item = impl Into<LinearTerm>>> From<I> for LinearExpr {
    fn from(iter: I) -> Self {
        let mut expr = LinearExpr::empty();
        for term in iter {
            let LinearTerm(var, coeff) = term.into();
            expr.add(var, coeff);
        }
        expr
    }
}

impl std::iter::FromIterator<(Variable, f64)> for LinearExpr {
    fn from_iter<I: IntoIterator<Item = (Variable, f64)>>(iter: I) -> Self {
        let mut expr = LinearExpr::empty();
        for (var, coeff) in iter {
            expr.add(var, coeff);
        }
        expr
    }
}

impl std::iter::Extend<(Variable, f64)> for LinearExpr {
    fn extend<I: IntoIterator<Item = (Variable, f64)>>(&mut self, iter: I) {
        for (var, coeff) in iter {
            self.add(var, coeff);
        }
    }
}

/// An operator specifying the relation between left-hand and right-hand sides of the constraint.  (`lib.rs:161-169`)
#[derive(Clone, Copy, Debug)]
pub enum ComparisonOp {
    /// The == operator (equal to)
    Eq,
    /// The <= operator (less than or equal to)
    Le,
    /// The >= operator (greater than or equal to)
    Ge,
}

impl ComparisonOp {
    fn to_c(self) -> std::os::raw::c_int {
        match self {
            ComparisonOp::Eq => sys::MLP_EQ,
            ComparisonOp::Le => sys::MLP_LE,
            ComparisonOp::Ge => sys::MLP_GE,
        }
    }
}

/// An error encountered while solving a problem.  (`lib.rs:172-190`)
#[derive(Clone, Debug, PartialEq)]
pub enum Error {
    /// Constrains can't simultaneously be satisfied.
    Infeasible,
    /// The objective function is unbounded.
    Unbounded,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        let msg = match self {
            Error::Infeasible => "problem is infeasible",
            Error::Unbounded => "problem is unbounded",
        };
        msg.fmt(f)
    }
}

impl std::error::Error for Error {}

fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::mlp_last_error()) }.to_string_lossy().into_owned()
}

/// Status of a C-ABI call -> the reference's behaviour: 0 Ok, 1 / 2 the two `Error`s, negative values are the
/// reference's panics (duplicate variable in an expression, out-of-range `Variable`, Gomory cut on a non-basic
/// variable, singular basis) or a HIP-level failure.
fn check(status: std::os::raw::c_int) -> Result<(), Error> {
    match status {
        sys::MLP_OK => Ok(()),
        sys::MLP_INFEASIBLE => Err(Error::Infeasible),
        sys::MLP_UNBOUNDED => Err(Error::Unbounded),
        code => panic!("minilp (HIP engine) error {}: {}", code, last_error()),
    }
}

/// A specification of a linear programming problem.  (`lib.rs:193-311`)
#[derive(Clone)]
pub struct Problem {
    direction: OptimizationDirection,
    obj_coeffs: Vec<f64>,
    var_mins: Vec<f64>,
    var_maxs: Vec<f64>,
    // rows in CSR form, ready for one bulk FFI crossing at solve()
    row_ptr: Vec<u64>,
    row_vars: Vec<u32>,
    row_coeffs: Vec<f64>,
    row_ops: Vec<i32>,
    row_rhs: Vec<f64>,
}

impl std::fmt::Debug for Problem {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        // (only printing lengths here, like the reference)
        f.debug_struct("Problem")
            .field("direction", &self.direction)
            .field("num_vars", &self.obj_coeffs.len())
            .field("num_constraints", &self.row_rhs.len())
            .finish()
    }
}

impl Problem {
    /// Create a new problem instance.
    pub fn new(direction: OptimizationDirection) -> Self {
        Problem {
            direction,
            obj_coeffs: vec![],
            var_mins: vec![],
            var_maxs: vec![],
            row_ptr: vec![0],
            row_vars: vec![],
            row_coeffs: vec![],
            row_ops: vec![],
            row_rhs: vec![],
        }
    }

    /// Add a new variable to the problem.  `obj_coeff` is its coefficient in the objective, `min` and `max`
    /// may be infinite.  (`lib.rs:233-242`; the library negates the objective for `Maximize` like `lib.rs:235-238`.)
    pub fn add_var(&mut self, obj_coeff: f64, (min, max): (f64, f64)) -> Variable {
        let var = Variable(self.obj_coeffs.len());
        self.obj_coeffs.push(obj_coeff);
        self.var_mins.push(min);
        self.var_maxs.push(max);
        var
    }

    /// Add a linear constraint to the problem.
    ///
    /// # Panics
    ///
    /// Will panic if a variable was added more than once to the left-hand side expression
    /// (`lib.rs:276-289`: `CsVec::new` on unsorted / duplicate indices).
    pub fn add_constraint(&mut self, expr: impl Into<LinearExpr>, cmp_op: ComparisonOp, rhs: f64) {
        let expr = expr.into();
        let mut seen = std::collections::HashSet::with_capacity(expr.vars.len());
        for &v in &expr.vars {
            assert!(v < self.obj_coeffs.len(), "variable {} is not part of this problem", v);
            assert!(seen.insert(v), "variable {} was added more than once to the expression", v);
        }
        self.row_vars.extend(expr.vars.iter().map(|&v| v as u32));
        self.row_coeffs.extend_from_slice(&expr.coeffs);
        self.row_ptr.push(self.row_vars.len() as u64);
        self.row_ops.push(cmp_op.to_c());
        self.row_rhs.push(rhs);
    }

    /// The C-side model: one `mlp_problem` filled with two bulk calls (freed by the caller).
    pub(crate) fn to_c(&self) -> *mut sys::mlp_problem {
        let dir = match self.direction {
            OptimizationDirection::Minimize => sys::MLP_MINIMIZE,
            OptimizationDirection::Maximize => sys::MLP_MAXIMIZE,
        };
        unsafe {
            let p = sys::mlp_problem_new(dir);
            assert!(!p.is_null(), "mlp_problem_new failed: {}", last_error());
            let st = sys::mlp_problem_add_vars(
                p,
                self.obj_coeffs.len() as u64,
                self.obj_coeffs.as_ptr(),
                self.var_mins.as_ptr(),
                self.var_maxs.as_ptr(),
            );
            assert_eq!(st, sys::MLP_OK, "{}", last_error());
            let st = sys::mlp_problem_add_constraints_csr(
                p,
                self.row_rhs.len() as u64,
                self.row_ptr.as_ptr(),
                self.row_vars.as_ptr(),
                self.row_coeffs.as_ptr(),
                self.row_ops.as_ptr(),
                self.row_rhs.as_ptr(),
            );
            assert_eq!(st, sys::MLP_OK, "{}", last_error());
            p
        }
    }

    /// Solve the problem, finding the optimal objective function value and variable values.
    ///
    /// # Errors
    ///
    /// Will return an error, if the problem is infeasible (constraints can't be satisfied)
    /// or if the objective value is unbounded.  (`lib.rs:291-310`)
    pub fn solve(&self) -> Result<Solution, Error> {
        let p = self.to_c();
        let mut s: *mut sys::mlp_solution = ptr::null_mut();
        let st = unsafe { sys::mlp_problem_solve(p, &mut s) };
        unsafe { sys::mlp_problem_free(p) };
        check(st)?;
        Ok(Solution::from_raw(s, self.obj_coeffs.len()))
    }
}

/// A solution of a problem: optimal objective function value and variable values.
///
/// Note that a `Solution` instance contains the whole solver machinery which can require
/// a lot of memory for larger problems (here: device memory).  (`lib.rs:313-424`)
pub struct Solution {
    raw: *mut sys::mlp_solution,
    num_vars: usize,
    // `var_value` and `Index` hand out `&f64` (lib.rs:344, 426): the values are copied from the device once per
    // solved state and cached; every mutator consumes `self`, so the cache can never be stale
    values: RefCell<Option<Box<[f64]>>>,
}

// The handle is used by one thread at a time (each owns its HIP stream and device buffers).
unsafe impl Send for Solution {}

impl Clone for Solution {
    /// `#[derive(Clone)]` of the reference (`lib.rs:313`): an independent deep copy of the solver state.
    fn clone(&self) -> Self {
        let raw = unsafe { sys::mlp_solution_clone(self.raw) };
        assert!(!raw.is_null(), "mlp_solution_clone failed: {}", last_error());
        Solution::from_raw(raw, self.num_vars)
    }
}

impl Drop for Solution {
    fn drop(&mut self) {
        if !self.raw.is_null() {
            unsafe { sys::mlp_solution_free(self.raw) };
            self.raw = ptr::null_mut();
        }
    }
}

impl std::fmt::Debug for Solution {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        // (only printing lengths here, like the reference)
        f.debug_struct("Solution").field("num_vars", &self.num_vars).finish()
    }
}

impl Solution {
    fn from_raw(raw: *mut sys::mlp_solution, num_vars: usize) -> Self {
        Solution { raw, num_vars, values: RefCell::new(None) }
    }

    fn values(&self) -> &[f64] {
        if self.values.borrow().is_none() {
            let mut buf = vec![0.0f64; self.num_vars].into_boxed_slice();
            let st = unsafe { sys::mlp_solution_values(self.raw, buf.as_mut_ptr(), self.num_vars as u32) };
            assert_eq!(st, sys::MLP_OK, "{}", last_error());
            *self.values.borrow_mut() = Some(buf);
        }
        // SAFETY: once filled the box is never replaced or dropped while `self` lives (mutators take `self` by value)
        let b = self.values.borrow();
        let s: &[f64] = b.as_ref().unwrap();
        unsafe { std::slice::from_raw_parts(s.as_ptr(), s.len()) }
    }

    /// Optimal value of the objective function.
    pub fn objective(&self) -> f64 {
        unsafe { sys::mlp_solution_objective(self.raw) }
    }

    /// Value of the variable at optimum.  Note that you can use indexing operations to get variable values.
    pub fn var_value(&self, var: Variable) -> &f64 {
        assert!(var.0 < self.num_vars);
        &self.values()[var.0]
    }

    /// Iterate over the variable-value pairs of the solution.
    pub fn iter(&self) -> SolutionIter {
        SolutionIter { solution: self, var_idx: 0 }
    }

    /// Runs one consuming C call: on a non-zero status the library has already freed the solver
    /// (consume-on-error, the reference drops `self` at `lib.rs:359, 385`).
    fn consume(mut self, f: impl FnOnce(*mut *mut sys::mlp_solution) -> std::os::raw::c_int) -> Result<Self, Error> {
        let mut raw = self.raw;
        self.raw = ptr::null_mut(); // whatever happens, `self` no longer owns it
        let st = f(&mut raw);
        check(st)?; // on Err the library freed and nulled `raw`
        Ok(Solution::from_raw(raw, self.num_vars))
    }

    /// Add another constraint and return the solution to the updated problem.  (`lib.rs:368-388`)
    ///
    /// # Errors
    ///
    /// Will return an error if the problem becomes infeasible with the additional constraint.
    pub fn add_constraint(self, expr: impl Into<LinearExpr>, cmp_op: ComparisonOp, rhs: f64) -> Result<Self, Error> {
        let expr = expr.into();
        let vars: Vec<u32> = expr.vars.iter().map(|&v| v as u32).collect();
        self.consume(|s| unsafe {
            sys::mlp_solution_add_constraint(s, vars.as_ptr(), expr.coeffs.as_ptr(), vars.len() as u64, cmp_op.to_c(), rhs)
        })
    }

    /// Fix the variable to the specified value and return the solution to the updated problem.  (`lib.rs:390-397`)
    ///
    /// # Errors
    ///
    /// Will return an error if the problem becomes infeasible with the additional constraint.
    pub fn fix_var(self, var: Variable, val: f64) -> Result<Self, Error> {
        assert!(var.0 < self.num_vars);
        self.consume(|s| unsafe { sys::mlp_solution_fix_var(s, var.0 as u32, val) })
    }

    /// If the variable was fixed with [`fix_var`](Self::fix_var) before, remove that constraint and return the
    /// solution to the updated problem and a boolean indicating if the variable was really fixed.  (`lib.rs:399-417`)
    pub fn unfix_var(self, var: Variable) -> (Self, bool) {
        assert!(var.0 < self.num_vars);
        let mut was: std::os::raw::c_int = 0;
        let sol = self
            .consume(|s| unsafe { sys::mlp_solution_unfix_var(s, var.0 as u32, &mut was) })
            .expect("unfix_var cannot make the problem infeasible or unbounded (lib.rs:433 unwrap)");
        (sol, was != 0)
    }

    /// Add a Gomory cut constraint to the problem and return the solution.  (`lib.rs:419-423`)
    ///
    /// # Errors
    ///
    /// Will return an error if the problem becomes infeasible with the additional constraint.
    ///
    /// # Panics
    ///
    /// Will panic if the variable is not basic (variable is basic if it has value other than its bounds).
    pub fn add_gomory_cut(self, var: Variable) -> Result<Self, Error> {
        assert!(var.0 < self.num_vars);
        self.consume(|s| unsafe { sys::mlp_solution_add_gomory_cut(s, var.0 as u32) })
    }

    /// The raw handle, for the engine-level stepping API and the diagnostics of `minilp-hip-sys`
    /// (not part of the reference's API).
    pub fn as_raw(&self) -> *mut sys::mlp_solution {
        self.raw
    }
}

impl std::ops::Index<Variable> for Solution {
    type Output = f64;

    fn index(&self, var: Variable) -> &Self::Output {
        self.var_value(var)
    }
}

/// An iterator over the variable-value pairs of a [`Solution`].  (`lib.rs:435-462`)
#[derive(Debug, Clone)]
pub struct SolutionIter<'a> {
    solution: &'a Solution,
    var_idx: usize,
}

impl<'a> Iterator for SolutionIter<'a> {
    type Item = (Variable, &'a f64);

    fn next(&mut self) -> Option<Self::Item> {
        if self.var_idx < self.solution.num_vars {
            let var_idx = self.var_idx;
            self.var_idx += 1;
            Some((Variable(var_idx), &self.solution.values()[var_idx]))
        } else {
            None
        }
    }
}

impl<'a> IntoIterator for &'a Solution {
    type Item = (Variable, &'a f64);
    type IntoIter = SolutionIter<'a>;

    fn into_iter(self) -> Self::IntoIter {
        self.iter()
    }
}

#[cfg(test)]
mod tests {
    //! The reference's own tests of `lib.rs:466-646`, unchanged in substance: they are the acceptance tests of a
    //! drop-in.  (The same cases run on the GPU box through the ctypes mirror: tests/test_hip_parity.py.)
    use super::*;

    #[test]
    fn optimize() {
        let mut problem = Problem::new(OptimizationDirection::Maximize);
        let v1 = problem.add_var(3.0, (12.0, f64::INFINITY));
        let v2 = problem.add_var(4.0, (5.0, f64::INFINITY));
        problem.add_constraint(&[(v1, 1.0), (v2, 1.0)], ComparisonOp::Le, 20.0);
        problem.add_constraint(&[(v2, -4.0), (v1, 1.0)], ComparisonOp::Ge, -20.0);
        let sol = problem.solve().unwrap();
        assert_eq!(sol[v1], 12.0);
        assert_eq!(sol[v2], 8.0);
        assert_eq!(sol.objective(), 68.0);
    }

    #[test]
    fn fix_unfix_var() {
        let mut problem = Problem::new(OptimizationDirection::Maximize);
        let v1 = problem.add_var(1.0, (0.0, 3.0));
        let v2 = problem.add_var(2.0, (0.0, 3.0));
        problem.add_constraint(&[(v1, 1.0), (v2, 1.0)], ComparisonOp::Le, 4.0);
        problem.add_constraint(&[(v1, 1.0), (v2, 1.0)], ComparisonOp::Ge, 1.0);
        let orig_sol = problem.solve().unwrap();
        let sol = orig_sol.clone().fix_var(v1, 0.5).unwrap();
        assert_eq!(sol[v1], 0.5);
        assert_eq!(sol[v2], 3.0);
        assert_eq!(sol.objective(), 6.5);
        let (sol, was_fixed) = sol.unfix_var(v1);
        assert!(was_fixed);
        assert_eq!(sol[v1], 1.0);
        assert_eq!(sol.objective(), 7.0);
    }

    #[test]
    fn infeasible_constraint_consumes_the_solution() {
        let mut problem = Problem::new(OptimizationDirection::Minimize);
        let v = problem.add_var(1.0, (0.0, 1.0));
        let sol = problem.solve().unwrap();
        assert_eq!(sol.add_constraint(&[(v, 1.0)], ComparisonOp::Ge, 2.0).unwrap_err(), Error::Infeasible);
    }
}
