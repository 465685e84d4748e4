// mps.cpp — free-format MPS reader of the product (host side; no arithmetic).  Accepts what
// MpsFile::parse accepts (mps.rs:39-328): sections NAME / ROWS / COLUMNS / RHS / [RANGES] /
// [BOUNDS] / ENDATA, '*' comments, first RHS/RANGES/BOUNDS vector only, bound types UP LO FX FR,
// negative UP without LO => (-inf, ub] (mps.rs:299), a RANGES entry => a >= row and a <= row
// (mps.rs:306-321).  Errors carry the line number like the reference's io::Error text.
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <sstream>
#include <unordered_map>
#include <unordered_set>

#include "engine.h"

namespace mlp {
namespace {
struct Row {
    std::vector<uint32_t> vars;
    std::vector<double> coefs;
    int op;
    double rhs = 0.0, range = 0.0;
};
struct Col {
    bool has_lo = false, has_hi = false;
    double lo = 0.0, hi = 0.0, cost = 0.0;
};
[[noreturn]] void fail(size_t line, const std::string& msg) {
    throw MlpError(-1, "line " + std::to_string(line) + ": " + msg);
}
double to_f64(const std::string& s, size_t line) {
    // Locale-independent like the reference's f64::from_str (mps.rs:330-336): std::from_chars ignores LC_NUMERIC
    // (strtod would read "1,5" under a comma locale and stop at the '.' of "1.5") and rejects hex floats.
    const char* b = s.c_str();
    const char* e = b + s.size();
    if (b != e && *b == '+') {  // from_str accepts one leading '+', from_chars does not
        ++b;
        if (b != e && (*b == '+' || *b == '-')) fail(line, "couldn't parse float from string: `" + s + "`");
    }
    double v = 0.0;
    const std::from_chars_result r = std::from_chars(b, e, v, std::chars_format::general);
    if (r.ptr != e || b == e) fail(line, "couldn't parse float from string: `" + s + "`");
    if (r.ec == std::errc::result_out_of_range) {  // from_str saturates: 1e999 -> inf, 1e-999 -> 0
        const bool neg = *b == '-';
        // out of range either way: the decimal exponent of the leading non-zero digit decides (a 400-digit integer
        // mantissa with "e-5" overflows; "0.000...1e5" may still underflow)
        const char* q = b + ((*b == '-') ? 1 : 0);
        long lead = 0;          // decimal exponent of the first non-zero mantissa digit (10^lead), before the exponent part
        long int_digits = 0, frac_zeros = 0;
        bool seen_point = false, seen_nonzero = false;
        for (; q != e && *q != 'e' && *q != 'E'; ++q) {
            if (*q == '.') { seen_point = true; continue; }
            if (!seen_nonzero) {
                if (*q == '0') { if (seen_point) frac_zeros += 1; continue; }
                seen_nonzero = true;
                if (seen_point) lead = -(frac_zeros + 1);
            }
            if (!seen_point) int_digits += 1;
        }
        if (seen_nonzero && lead == 0) lead = int_digits - 1;
        long ex = 0;
        if (q != e) {  // exponent part (clamped: only its sign and rough size matter here)
            ++q;
            bool eneg = false;
            if (q != e && (*q == '+' || *q == '-')) { eneg = *q == '-'; ++q; }
            for (; q != e && *q >= '0' && *q <= '9'; ++q) ex = ex < 100000 ? ex * 10 + (*q - '0') : ex;
            if (eneg) ex = -ex;
        }
        const bool overflow = seen_nonzero && lead + ex > 0;
        v = overflow ? std::numeric_limits<double>::infinity() : 0.0;
        if (neg) v = -v;
    } else if (r.ec != std::errc()) {
        fail(line, "couldn't parse float from string: `" + s + "`");
    }
    return v;
}
}  // namespace

MpsData parse_mps(const std::string& text, int direction) {
    const double INF = std::numeric_limits<double>::infinity();
    enum Section { S_START, S_NAME_DONE, S_ROWS, S_COLUMNS, S_RHS, S_RANGES, S_BOUNDS, S_END };
    Section sec = S_START;
    MpsData out;
    std::string obj_name;
    bool have_obj = false;
    std::unordered_set<std::string> free_rows;
    std::unordered_map<std::string, size_t> row_idx, col_idx;
    std::vector<Row> rows;
    std::vector<Col> cols;
    std::string vec_name[3];  // first RHS / RANGES / BOUNDS vector
    bool have_vec[3] = {false, false, false};

    std::istringstream in(text);
    std::string line;
    size_t lineno = 0;
    std::vector<std::string> tok;
    auto pairs_from = [&](size_t from, auto&& fn) {  // one or two (name, number) pairs (mps.rs:402-433)
        if (tok.size() < from + 2) fail(lineno, "unexpected end of line");
        fn(tok[from], to_f64(tok[from + 1], lineno));
        if (tok.size() > from + 2) {
            if (tok.size() < from + 4) fail(lineno, "unexpected end of line");
            fn(tok[from + 2], to_f64(tok[from + 3], lineno));
        }
    };
    while (std::getline(in, line)) {
        lineno += 1;
        if (!line.empty() && line[0] == '*') continue;
        size_t end = line.find_last_not_of(" \t\r\n\f\v");
        if (end == std::string::npos) continue;
        line.resize(end + 1);
        tok.clear();
        {
            std::istringstream ls(line);
            std::string w;
            while (ls >> w) tok.push_back(w);
        }
        const bool data_line = line[0] == ' ';
        if (!data_line) {  // section header
            if (sec == S_START) {
                if (tok[0] != "NAME") fail(lineno, "expected NAME section");
                out.name = tok.size() > 1 ? tok[1] : "";
                sec = S_NAME_DONE;
            } else if (sec == S_NAME_DONE) {
                if (line != "ROWS") fail(lineno, "expected ROWS section");
                sec = S_ROWS;
            } else if (sec == S_ROWS) {
                if (!have_obj) fail(lineno, "objective function name not declared");
                if (line != "COLUMNS") fail(lineno, "expected COLUMNS section");
                sec = S_COLUMNS;
            } else if (sec == S_COLUMNS) {
                if (line != "RHS") fail(lineno, "expected RHS section");
                sec = S_RHS;
            } else if (sec == S_RHS && line == "RANGES") {
                sec = S_RANGES;
            } else if ((sec == S_RHS || sec == S_RANGES) && line == "BOUNDS") {
                sec = S_BOUNDS;
            } else if ((sec == S_RHS || sec == S_RANGES || sec == S_BOUNDS) && line == "ENDATA") {
                sec = S_END;
                break;
            } else {
                fail(lineno, "expected ENDATA section");
            }
            continue;
        }
        switch (sec) {
            case S_ROWS: {
                if (tok.size() < 2) fail(lineno, "unexpected end of line");
                const std::string &ty = tok[0], &name = tok[1];
                if (ty == "N") {
                    if (!have_obj) { have_obj = true; obj_name = name; }
                    else free_rows.insert(name);
                    break;
                }
                Row r;
                if (ty == "L") r.op = 1;
                else if (ty == "G") r.op = 2;
                else if (ty == "E") r.op = 0;
                else fail(lineno, "unexpected row type " + ty);
                if (!row_idx.emplace(name, rows.size()).second) fail(lineno, "row " + name + " already declared");
                rows.push_back(r);
                break;
            }
            case S_COLUMNS: {
                const std::string& name = tok[0];
                if (out.var_names.empty() || out.var_names.back() != name) {
                    if (col_idx.count(name)) fail(lineno, "variable " + name + " already declared");
                    col_idx.emplace(name, cols.size());
                    out.var_names.push_back(name);
                    cols.push_back(Col());
                }
                uint32_t j = (uint32_t)(cols.size() - 1);
                pairs_from(1, [&](const std::string& key, double val) {
                    if (key == obj_name) cols[j].cost = val;
                    else {
                        auto it = row_idx.find(key);
                        if (it != row_idx.end()) {
                            rows[it->second].vars.push_back(j);
                            rows[it->second].coefs.push_back(val);
                        } else if (!free_rows.count(key)) fail(lineno, "unknown constraint: " + key);
                    }
                });
                break;
            }
            case S_RHS:
            case S_RANGES: {
                int w = sec == S_RHS ? 0 : 1;
                if (!have_vec[w]) { have_vec[w] = true; vec_name[w] = tok[0]; }
                else if (vec_name[w] != tok[0]) break;
                pairs_from(1, [&](const std::string& key, double val) {
                    if (w == 0 && key == obj_name) fail(lineno, "setting objective in RHS section is not supported");
                    auto it = row_idx.find(key);
                    if (it == row_idx.end()) fail(lineno, "unknown constraint: " + key);
                    (w == 0 ? rows[it->second].rhs : rows[it->second].range) = val;
                });
                break;
            }
            case S_BOUNDS: {
                if (tok.size() < 2) fail(lineno, "unexpected end of line");
                if (!have_vec[2]) { have_vec[2] = true; vec_name[2] = tok[1]; }
                else if (vec_name[2] != tok[1]) break;
                if (tok.size() < 3) fail(lineno, "unexpected end of line");
                auto it = col_idx.find(tok[2]);
                if (it == col_idx.end()) fail(lineno, "unknown variable: " + tok[2]);
                Col& c = cols[it->second];
                const std::string& ty = tok[0];
                if (ty == "FR") {
                    c.has_lo = c.has_hi = true;
                    c.lo = -INF;
                    c.hi = INF;
                    break;
                }
                if (tok.size() < 4) fail(lineno, "unexpected end of line");
                double val = to_f64(tok[3], lineno);
                if (ty == "LO") { c.has_lo = true; c.lo = val; }
                else if (ty == "UP") { c.has_hi = true; c.hi = val; }
                else if (ty == "FX") { c.has_lo = c.has_hi = true; c.lo = c.hi = val; }
                else fail(lineno, "bound type " + ty + " is not supported");
                break;
            }
            default:
                fail(lineno, "data line outside of a section");
        }
    }
    if (sec != S_END) fail(lineno + 1, sec == S_START ? "expected NAME section" : "expected ENDATA section");

    out.problem.direction = direction;
    for (const Col& c : cols) {  // mps.rs:293-304
        double lo, hi;
        if (c.has_lo && c.has_hi) { lo = c.lo; hi = c.hi; }
        else if (c.has_lo) { lo = c.lo; hi = INF; }
        else if (c.has_hi && c.hi < 0.0) { lo = -INF; hi = c.hi; }
        else if (c.has_hi) { lo = 0.0; hi = c.hi; }
        else { lo = 0.0; hi = INF; }
        out.problem.add_var(c.cost, lo, hi);
    }
    for (const Row& r : rows) {  // mps.rs:306-321
        if (r.range == 0.0) {
            out.problem.add_constraint(r.vars.data(), r.coefs.data(), r.vars.size(), r.op, r.rhs);
        } else {
            double lo, hi;
            if (r.op == 2) { lo = r.rhs; hi = r.rhs + std::fabs(r.range); }
            else if (r.op == 1) { lo = r.rhs - std::fabs(r.range); hi = r.rhs; }
            else if (r.range > 0.0) { lo = r.rhs; hi = r.rhs + r.range; }
            else { lo = r.rhs + r.range; hi = r.rhs; }
            out.problem.add_constraint(r.vars.data(), r.coefs.data(), r.vars.size(), 2, lo);
            out.problem.add_constraint(r.vars.data(), r.coefs.data(), r.vars.size(), 1, hi);
        }
    }
    return out;
}
}  // namespace mlp
