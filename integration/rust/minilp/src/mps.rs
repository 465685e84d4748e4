//! `MpsFile` (`src/mps.rs:7-16, 39`) over the library's MPS reader (`mlp_mps_parse`, csrc/mps.cpp): free-format
//! tokens, first RHS / RANGES / BOUNDS vector only, negative-UP-without-LO rule, RANGES -> two rows — the rules of
//! `mps.rs:39-431`, pinned against the reference's `test_parse_mps_file` (obj 54) in tests/test_mps_tsp.py.
use crate::{last_error, OptimizationDirection, Problem, Variable};
use minilp_hip_sys as sys;
use std::collections::HashMap;
use std::ffi::CStr;
use std::io;

/// A linear programming problem parsed from an MPS file.
#[derive(Clone)]
pub struct MpsFile {
    /// Value of the NAME section.
    pub problem_name: String,
    /// Variables by their names in the COLUMNS section.
    pub variables: HashMap<String, Variable>,
    /// The parsed problem.
    pub problem: Problem,
}

impl std::fmt::Debug for MpsFile {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("MpsFile")
            .field("problem_name", &self.problem_name)
            .field("num_vars", &self.variables.len())
            .finish()
    }
}

impl MpsFile {
    /// Parses a linear programming problem from an MPS file.  (`mps.rs:39`)
    ///
    /// # Errors
    ///
    /// `io::ErrorKind::InvalidData` with the `line N: ...` message for syntax errors, I/O errors of the reader as is.
    pub fn parse<R: io::BufRead>(mut input: R, direction: OptimizationDirection) -> io::Result<Self> {
        let mut text = Vec::new();
        input.read_to_end(&mut text)?;
        let dir = match direction {
            OptimizationDirection::Minimize => sys::MLP_MINIMIZE,
            OptimizationDirection::Maximize => sys::MLP_MAXIMIZE,
        };
        let mut f: *mut sys::mlp_mps = std::ptr::null_mut();
        let st = unsafe { sys::mlp_mps_parse(text.as_ptr() as *const _, text.len() as u64, dir, &mut f) };
        if st != sys::MLP_OK {
            return Err(io::Error::new(io::ErrorKind::InvalidData, last_error()));
        }
        // copy the parsed model into the plain-data `Problem` of this crate, then free the C-side objects
        let file = unsafe {
            let problem_name = CStr::from_ptr(sys::mlp_mps_name(f)).to_string_lossy().into_owned();
            let n = sys::mlp_mps_num_vars(f);
            let cp = sys::mlp_mps_problem(f); // a clone owned by us
            let mut problem = Problem::new(direction);
            let mut variables = HashMap::with_capacity(n as usize);
            for i in 0..n {
                let (mut obj, mut lo, mut hi) = (0.0, 0.0, 0.0);
                sys::mlp_problem_var(cp, i, &mut obj, &mut lo, &mut hi);
                let var = problem.add_var(obj, (lo, hi)); // (obj as the user wrote it: un-negated)
                let name = CStr::from_ptr(sys::mlp_mps_var_name(f, i)).to_string_lossy().into_owned();
                variables.insert(name, var);
            }
            let m = sys::mlp_problem_num_constraints(cp);
            for r in 0..m {
                let (mut op, mut rhs) = (0, 0.0);
                let k = sys::mlp_problem_constraint(cp, r, std::ptr::null_mut(), std::ptr::null_mut(), 0, &mut op, &mut rhs);
                let mut vars = vec![0u32; k as usize];
                let mut coeffs = vec![0.0f64; k as usize];
                sys::mlp_problem_constraint(cp, r, vars.as_mut_ptr(), coeffs.as_mut_ptr(), k, &mut op, &mut rhs);
                let expr: crate::LinearExpr = vars.iter().zip(&coeffs).map(|(&v, &c)| (Variable(v as usize), c)).collect();
                let cmp = match op {
                    sys::MLP_EQ => crate::ComparisonOp::Eq,
                    sys::MLP_LE => crate::ComparisonOp::Le,
                    _ => crate::ComparisonOp::Ge,
                };
                problem.add_constraint(expr, cmp, rhs);
            }
            sys::mlp_problem_free(cp);
            sys::mlp_mps_free(f);
            MpsFile { problem_name, variables, problem }
        };
        Ok(file)
    }
}
