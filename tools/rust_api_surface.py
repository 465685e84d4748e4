#!/usr/bin/env python3
"""Makes the Rust drop-in crate checkable WITHOUT a compiler.

Extracts every `pub` item signature (fn / struct / enum / trait / type / const / impl-block methods) from the reference's
`src/lib.rs` and `src/mps.rs` and from `integration/rust/minilp/src/{lib,mps}.rs`, normalises whitespace, and writes
`integration/rust/API_SURFACE.md`: the two lists side by side, items only in the reference (MISSING from the crate), items only
in the crate (extensions), and signatures that differ.  Run in the build container (where /root/reference exists); the
output is committed.  `--check` exits non-zero when an item of the reference is missing or differs."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MINILP_REFERENCE", "/root/reference")


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return "\n".join(ln.split("//")[0] for ln in src.splitlines())


def pub_items(path):
    """{key: normalised signature} of the public surface: `pub fn` (free and in `impl` blocks, keyed by Type::name), `pub struct`,
    `pub enum`, `pub type`, `pub const`, and `impl Trait for Type` headers (the operator / conversion surface)."""
    src = strip_comments(open(path).read())
    items = {}
    # split off #[cfg(test)] modules
    src = re.split(r"#\[cfg\(test\)\]", src)[0]
    # derived traits are part of the surface: `#[derive(Clone, Debug)] pub struct T` == `impl Clone for T`, `impl Debug for T`
    for md in re.finditer(r"#\[derive\(([^)]*)\)\]\s*(?:#\[[^\]]*\]\s*)*pub\s+(?:struct|enum)\s+(\w+)", src):
        for tr in md.group(1).split(","):
            tr = tr.strip()
            if tr:
                items["impl %s for %s" % (tr, md.group(2))] = "impl %s for %s  (derived or written out)" % (tr, md.group(2))
    impl_stack = []   # (brace depth at which the impl body opened, type name, is_pub_trait_impl)
    depth = 0
    i = 0
    tokens = re.finditer(r"\bimpl\b[^{;]*\{|\bpub\s+(?:fn|struct|enum|type|const|trait)\b[^{;]*[{;]|[{}]", src)
    for mt in tokens:
        t = mt.group(0)
        if t == "{":
            depth += 1
        elif t == "}":
            depth -= 1
            while impl_stack and impl_stack[-1][0] > depth:
                impl_stack.pop()
        elif t.startswith("impl"):
            head = " ".join(t[:-1].split())
            m2 = re.match(r"impl(?:<[^>]*>)?\s+(?:(.+?)\s+for\s+)?([A-Za-z_][\w:<>', ]*)", head)
            trait, typ = (m2.group(1), m2.group(2).strip()) if m2 else (None, head)
            typ = re.sub(r"<.*", "", typ).strip()
            depth += 1
            impl_stack.append((depth, typ))
            if trait:
                tshort = " ".join(trait.split())
                if tshort in ("Clone", "Debug", "Copy", "PartialEq", "Eq", "Hash", "Default", "PartialOrd", "Ord"):
                    items["impl %s for %s" % (tshort, typ)] = "impl %s for %s  (derived or written out)" % (tshort, typ)
                else:
                    items["impl %s for %s" % (tshort, typ)] = head
        else:
            sig = " ".join(t[:-1].split())
            sig = re.sub(r"\(\s*", "(", sig)
            sig = re.sub(r",?\s*\)", ")", sig)
            sig = re.sub(r"(?<![&*])\bmut (self\b|[a-z_]\w*\s*:(?!:))", r"\1", sig)   # a `mut` binding of a parameter is not part of the signature
            opens = t.endswith("{")
            kind = sig.split()[1]
            name = re.match(r"pub\s+\w+\s+([A-Za-z_]\w*)", sig).group(1)
            key = ("%s::%s" % (impl_stack[-1][1], name)) if (kind == "fn" and impl_stack) else "%s %s" % (kind, name)
            if kind in ("struct", "enum", "trait") and opens:
                # keep the body of a pub struct / enum (fields and variants are part of the surface)
                j, d = mt.end(), 1
                while d and j < len(src):
                    d += {"{": 1, "}": -1}.get(src[j], 0)
                    j += 1
                body = " ".join(src[mt.end():j - 1].split())
                if kind == "struct":  # private fields are not part of the public surface
                    fields, cur, dd = [], "", 0
                    for ch in body:   # split at top-level commas only (generic arguments contain commas)
                        dd += {"<": 1, "(": 1, ">": -1, ")": -1}.get(ch, 0)
                        if ch == "," and dd == 0:
                            fields.append(cur)
                            cur = ""
                        else:
                            cur += ch
                    fields.append(cur)
                    pubf = [f.strip() for f in fields if f.strip().startswith("pub ") and not f.strip().startswith("pub(crate)")]
                    body = ", ".join(pubf) if pubf else "/* private fields */"
                sig = sig + " { " + body + " }"
            items[key] = sig
            if opens:
                depth += 1
    return items


def main():
    pairs = [("lib.rs", os.path.join(REF, "src", "lib.rs"), os.path.join(ROOT, "integration", "rust", "minilp", "src", "lib.rs")),
             ("mps.rs", os.path.join(REF, "src", "mps.rs"), os.path.join(ROOT, "integration", "rust", "minilp", "src", "mps.rs"))]
    out = ["# Public API surface: reference crate `minilp` 0.2.2 vs the drop-in crate `integration/rust/minilp`", "",
           "Generated by `python tools/rust_api_surface.py` in the build container (the reference sources are read from `%s`; nothing of" % REF,
           "them is copied here beyond the one-line signatures a drop-in crate has to restate).  No Rust toolchain exists in the image, so",
           "this file — not a compiler — is what pins the drop-in claim: every `pub` item of the reference must appear in the crate with the",
           "same signature.  `tests/test_abi.py` re-checks the committed lists against the crate source and checks every `extern \"C\"`",
           "declaration of `minilp-hip-sys` against `libminilp_hip.so` and `include/minilp_hip.h`.", ""]
    bad = 0
    for name, ref_path, our_path in pairs:
        ref, ours = pub_items(ref_path), pub_items(our_path)
        missing = sorted(k for k in ref if k not in ours)
        extra = sorted(k for k in ours if k not in ref)
        differ = sorted(k for k in ref if k in ours and ref[k] != ours[k])
        same = sorted(k for k in ref if k in ours and ref[k] == ours[k])
        bad += len(missing) + len(differ)
        out += ["## `%s`" % name, "", "* public items in the reference: **%d**; present in the crate with an identical signature: **%d**; "
                "present with a different signature: **%d**; missing: **%d**; extensions of the crate: %d" % (len(ref), len(same), len(differ), len(missing), len(extra)), ""]
        out += ["### identical", ""] + ["* `%s`" % ref[k] for k in same] + [""]
        if differ:
            out += ["### different", ""]
            for k in differ:
                out += ["* reference: `%s`", "  crate:     `%s`"]
                out[-2] = out[-2] % ref[k]
                out[-1] = out[-1] % ours[k]
            out += [""]
        if missing:
            out += ["### MISSING from the crate", ""] + ["* `%s`" % ref[k] for k in missing] + [""]
        if extra:
            out += ["### extensions (not in the reference)", ""] + ["* `%s`" % ours[k] for k in extra] + [""]
    path = os.path.join(ROOT, "integration", "rust", "API_SURFACE.md")
    text = "\n".join(out) + "\n"
    if "--check" in sys.argv:
        print("missing or different items:", bad)
        sys.exit(1 if bad else 0)
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path, "| missing or different items:", bad)


if __name__ == "__main__":
    main()
