// Host-side helper of the TSP cutting-plane driver (examples/tsp.py; reference: examples/tsp.rs:437-539,
// Stoer-Wagner global minimum cut used to separate subtour-elimination constraints).  Not on the
// simplex path; it lives here so that the driver's separation step is native like the reference's.
#include <cstdint>
#include <cmath>
#include <vector>

extern "C" double mlp_util_min_cut(uint32_t n, const double* weights, uint8_t* side_out) {
    if (n == 0 || !weights || !side_out) return NAN;
    std::vector<double> w(weights, weights + (size_t)n * n);
    std::vector<std::vector<uint32_t>> groups(n);
    std::vector<uint32_t> active(n);
    for (uint32_t i = 0; i < n; ++i) {
        groups[i] = {i};
        active[i] = i;
        side_out[i] = 0;
    }
    double best_w = INFINITY;
    std::vector<uint32_t> best_set;
    std::vector<double> wt(n);
    std::vector<char> in_a(n);
    while (active.size() > 1) {
        const size_t na = active.size();
        const uint32_t a = active[0];
        for (size_t j = 0; j < na; ++j) {
            wt[j] = w[(size_t)a * n + active[j]];
            in_a[j] = 0;
        }
        in_a[0] = 1;
        uint32_t prev = a, last = a;
        double cut_of_phase = 0.0;
        for (size_t it = 1; it < na; ++it) {
            size_t j = 0;
            double bw = -INFINITY;
            bool found = false;
            for (size_t t = 0; t < na; ++t)  // most tightly connected vertex; first one on ties
                if (!in_a[t] && (!found || wt[t] > bw)) {
                    bw = wt[t];
                    j = t;
                    found = true;
                }
            prev = last;
            last = active[j];
            cut_of_phase = wt[j];
            in_a[j] = 1;
            const double* wl = &w[(size_t)last * n];
            for (size_t t = 0; t < na; ++t) wt[t] += wl[active[t]];
        }
        if (cut_of_phase < best_w) {
            best_w = cut_of_phase;
            best_set = groups[last];
        }
        groups[prev].insert(groups[prev].end(), groups[last].begin(), groups[last].end());
        for (uint32_t c = 0; c < n; ++c) w[(size_t)prev * n + c] += w[(size_t)last * n + c];
        for (uint32_t r = 0; r < n; ++r) w[(size_t)r * n + prev] += w[(size_t)r * n + last];
        w[(size_t)prev * n + prev] = 0.0;
        for (size_t t = 0; t < na; ++t)
            if (active[t] == last) {
                active.erase(active.begin() + (long)t);
                break;
            }
    }
    for (uint32_t v : best_set) side_out[v] = 1;
    return best_w;
}
