/* The README toy of the reference (README.md:31-44 / lib.rs:27-44) through the plain C ABI:
 *   maximize x + 2y  s.t.  x + y <= 4,  2x + y >= 2,  x >= 0,  0 <= y <= 3      ->  objective 7, x = 1, y = 3
 * then a warm-started extra constraint x <= 0.5 (Solution::add_constraint, lib.rs:368).
 *   gcc -I include examples/toy.c -L minilp_amd -lminilp_hip -Wl,-rpath,$PWD/minilp_amd -o toy && ./toy */
#include <math.h>
#include <stdio.h>

#include "minilp_hip.h"

int main(void) {
    mlp_problem* p = mlp_problem_new(MLP_MAXIMIZE);
    uint32_t x = mlp_problem_add_var(p, 1.0, 0.0, INFINITY);
    uint32_t y = mlp_problem_add_var(p, 2.0, 0.0, 3.0);
    uint32_t v[2] = {x, y};
    double c1[2] = {1.0, 1.0}, c2[2] = {2.0, 1.0};
    if (mlp_problem_add_constraint(p, v, c1, 2, MLP_LE, 4.0) != 0 || mlp_problem_add_constraint(p, v, c2, 2, MLP_GE, 2.0) != 0) {
        fprintf(stderr, "add_constraint: %s\n", mlp_last_error());
        return 2;
    }
    mlp_solution* s = NULL;
    int st = mlp_problem_solve(p, &s);
    if (st != 0) {
        fprintf(stderr, "solve: status %d %s\n", st, st < 0 ? mlp_last_error() : "");
        return 1;
    }
    double xv = 0.0, yv = 0.0;
    mlp_solution_var_value(s, x, &xv);
    mlp_solution_var_value(s, y, &yv);
    printf("objective %.12g x %.12g y %.12g\n", mlp_solution_objective(s), xv, yv);
    double half[1] = {1.0};
    st = mlp_solution_add_constraint(&s, &x, half, 1, MLP_LE, 0.5); /* consumes s on error, like the Rust receiver */
    if (st != 0) {
        fprintf(stderr, "add_constraint: status %d\n", st);
        return 1;
    }
    mlp_solution_var_value(s, x, &xv);
    mlp_solution_var_value(s, y, &yv);
    printf("warm-started objective %.12g x %.12g y %.12g\n", mlp_solution_objective(s), xv, yv);
    mlp_solution_free(s);
    mlp_problem_free(p);
    return 0;
}
