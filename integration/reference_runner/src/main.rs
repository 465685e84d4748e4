//! `reference_runner model.mps [--max]` — solve an MPS file with the real minilp 0.2.2 (same flow as the
//! reference's examples/solve_mps.rs) and print the objective and the wall time of `solve()`, excluding the
//! parse.  For pivot counts run with `RUST_LOG=debug` (the reference logs its iteration counter).
//! NOT COMPILED in the build environment of this repository (no Rust toolchain there).
use std::{fs::File, io::BufReader, time::Instant};

use minilp::{MpsFile, OptimizationDirection};

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() < 2 {
        eprintln!("usage: reference_runner model.mps [--max]");
        std::process::exit(2);
    }
    let direction = if args.iter().any(|a| a == "--max") {
        OptimizationDirection::Maximize
    } else {
        OptimizationDirection::Minimize
    };
    let t0 = Instant::now();
    let file = MpsFile::parse(BufReader::new(File::open(&args[1]).expect("open")), direction).expect("parse");
    let t_parse = t0.elapsed();
    let t1 = Instant::now();
    match file.problem.solve() {
        Ok(sol) => {
            let t_solve = t1.elapsed();
            println!(
                "problem {}: {} variables, parsed in {:.3}s, solved in {:.3}s",
                file.problem_name,
                file.variables.len(),
                t_parse.as_secs_f64(),
                t_solve.as_secs_f64()
            );
            println!("objective: {:.12}", sol.objective());
        }
        Err(e) => println!("problem {}: {}", file.problem_name, e),
    }
}
