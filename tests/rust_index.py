"""A small Rust tokenizer for the integration checks (tests/test_rust_identifiers.py): no Rust toolchain exists in this
image, so `integration/hip_backend.rs` and the hunks of `integration/lorikeet-hip.patch` have never met rustc.  What CAN be
checked mechanically is that every function, method, field, type and path they use from the reference tree exists there, and
is called with a number of arguments one of its definitions takes.

* `strip(text)`             comments, string / char literals and lifetimes removed (brackets stay balanced);
* `definitions(text)`       {name: set of arities} of every `fn` (arity without `self`), plus the names of struct / enum /
                            trait / type / const / static / mod items, struct fields and enum variants;
* `uses(text)`              calls `name(args)` / `.name(args)` / `Path::name(args)` with their top-level argument counts, field
                            accesses `.name`, and `Type::Name` paths.
Heuristic by design (macros, generics in expression position and closures are skipped over, not parsed) -- it reports what
it could not resolve, the test decides."""
import re

_IDENT = r"[A-Za-z_][A-Za-z0-9_]*"


def strip(text):
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            depth, i = 1, i + 2
            while i < n and depth:
                if text.startswith("/*", i):
                    depth, i = depth + 1, i + 2
                elif text.startswith("*/", i):
                    depth, i = depth - 1, i + 2
                else:
                    i += 1
        elif c == '"' or (c in "br" and re.match(r'b?r?#*"', text[i:i + 6]) and (i == 0 or not (text[i - 1].isalnum() or text[i - 1] == "_"))):
            m = re.match(r'(b?)(r?)(#*)"', text[i:])
            raw, hashes = m.group(2) == "r", m.group(3)
            i += m.end()
            close = '"' + hashes
            while i < n:
                if not raw and text[i] == "\\":
                    i += 2
                elif text.startswith(close, i):
                    i += len(close)
                    break
                else:
                    i += 1
            out.append('""')
        elif c == "'":
            m = re.match(r"'(\\.[^']*|[^'\\])'", text[i:])
            if m:                                   # a char literal
                out.append("' '")
                i += m.end()
            else:                                   # a lifetime: drop the quote, keep the name out of the way
                m = re.match(r"'" + _IDENT, text[i:])
                i += m.end() if m else 1
                out.append(" ")
        else:
            out.append(c)
            i += 1
    return "".join(out)


def _match(text, i, open_c, close_c):
    """index just past the bracket that closes text[i] == open_c (angle brackets: `->` and `=>` do not close)"""
    depth, j, n = 0, i, len(text)
    while j < n:
        ch = text[j]
        if ch == open_c:
            depth += 1
        elif ch == close_c and not (close_c == ">" and j > 0 and text[j - 1] in "-="):
            depth -= 1
            if depth == 0:
                return j + 1
        j += 1
    return n


def split_top(args):
    """top-level comma-separated pieces of an argument / parameter list (brackets of every kind respected; `<` only where it
    cannot be a comparison: after an identifier or `::`)"""
    parts, depth, cur, i, n = [], 0, [], 0, len(args)
    angle = 0
    while i < n:
        ch = args[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "<" and depth >= 0 and i > 0 and (args[i - 1].isalnum() or args[i - 1] in "_:") and not args.startswith("<=", i):
            # generics in a type position / turbofish; a comparison `a < b` has spaces around it in rustfmt'd code
            if i + 1 < n and args[i + 1] != " ":
                angle += 1
        elif ch == ">" and angle and not (i > 0 and args[i - 1] in "-="):
            angle -= 1
        if ch == "," and depth == 0 and angle == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
        i += 1
    last = "".join(cur).strip()
    if last:
        parts.append(last)
    return parts


def definitions(text):
    """of stripped text: (fns {name: set(arity)}, items set, fields set, variants set)"""
    fns, items, fields, variants = {}, set(), set(), set()
    for m in re.finditer(r"\bfn\s+(" + _IDENT + r")\s*", text):
        j = m.end()
        if j < len(text) and text[j] == "<":
            j = _match(text, j, "<", ">")
            while j < len(text) and text[j].isspace():
                j += 1
        if j >= len(text) or text[j] != "(":
            continue
        end = _match(text, j, "(", ")")
        params = split_top(text[j + 1:end - 1])
        params = [p for p in params if not re.match(r"^(&\s*)?(mut\s+)?self\b", p)]
        fns.setdefault(m.group(1), set()).add(len(params))
    for m in re.finditer(r"\b(struct|enum|trait|type|const|static|mod|union)\s+(?:mut\s+)?(" + _IDENT + r")", text):
        items.add(m.group(2))
    for m in re.finditer(r"\bstruct\s+" + _IDENT + r"\s*(?:<[^{;(]*>)?\s*(?:where[^{]*)?\{", text):
        end = _match(text, m.end() - 1, "{", "}")
        for p in split_top(text[m.end():end - 1]):
            f = re.match(r"(?:#\[[^\]]*\]\s*)*(?:pub(?:\([^)]*\))?\s+)?(" + _IDENT + r")\s*:", p)
            if f:
                fields.add(f.group(1))
    for m in re.finditer(r"\benum\s+" + _IDENT + r"\s*(?:<[^{]*>)?\s*\{", text):
        end = _match(text, m.end() - 1, "{", "}")
        for p in split_top(text[m.end():end - 1]):
            v = re.match(r"(?:#\[[^\]]*\]\s*)*(" + _IDENT + r")", p)
            if v:
                variants.add(v.group(1))
    return fns, items, fields, variants


_KEYWORDS = {"if", "while", "for", "match", "return", "fn", "let", "loop", "in", "as", "move", "unsafe", "else", "Some", "None", "Ok", "Err",
             "Box", "Vec", "String", "Self", "self", "super", "crate", "pub", "use", "impl", "where", "mut", "ref", "dyn", "struct", "enum"}


def uses(text):
    """of stripped text: calls [(kind, name, n_args)], kind in {"method", "path", "free"}; field accesses {name}; paths {(Type, Name)}"""
    calls, field_uses, paths = [], set(), set()
    for m in re.finditer(r"(\.|::)?\s*\b(" + _IDENT + r")\s*(::\s*<)?", text):
        name, j = m.group(2), m.end()
        if m.group(3):                              # turbofish: name::<T>(...)
            j = _match(text, j - 1, "<", ">")
        k = j
        while k < len(text) and text[k].isspace():
            k += 1
        lead = m.group(1)
        if k < len(text) and text[k] == "(" and not (k > 0 and text[k - 1] == "!"):
            if name in _KEYWORDS and lead is None:
                continue
            before = text[max(0, m.start() - 4):m.start()]
            if re.search(r"\bfn\s*$", text[max(0, m.start() - 8):m.start() + (1 if lead else 0)]):
                continue                           # a definition, not a call
            if text[j - 1:j] == "!" or text[m.end() - 1:m.end()] == "!":
                continue
            end = _match(text, k, "(", ")")
            n_args = len(split_top(text[k + 1:end - 1]))
            calls.append(("method" if lead == "." else "path" if lead == "::" else "free", name, n_args))
            del before
        elif lead == "." and not name[0].isdigit():
            if not (k < len(text) and text[k] == "!"):
                field_uses.add(name)
    for m in re.finditer(r"\b(" + _IDENT + r")\s*::\s*(" + _IDENT + r")\b", text):
        paths.add((m.group(1), m.group(2)))
    return calls, field_uses, paths
