"""A tiny proto3 front-end: ``.proto`` text -> ``FileDescriptorProto``.

``protoc`` / ``grpc_tools`` are not part of this toolchain, and the reference
generates its ``*_pb2`` modules at wheel-build time from files fetched over the
network (``src/python/library/build_wheel.py:125-153``).  Here the two protocol
files ship as text next to this module and are compiled when
``triton_client_amd.grpc`` is imported.

Supported subset (everything the two protocol files use): ``syntax``,
``package``, ``import``, ``message`` (nested), ``enum`` (nested), scalar /
message / enum fields, ``repeated``, ``map<K,V>``, ``oneof``, ``service`` with
unary and ``stream`` rpcs, ``//`` and ``/* */`` comments, ``option``/``reserved``
statements (skipped).
"""

from __future__ import annotations

import re

from google.protobuf import descriptor_pb2 as dpb

_F = dpb.FieldDescriptorProto

_SCALARS = {
    "double": _F.TYPE_DOUBLE,
    "float": _F.TYPE_FLOAT,
    "int64": _F.TYPE_INT64,
    "uint64": _F.TYPE_UINT64,
    "int32": _F.TYPE_INT32,
    "fixed64": _F.TYPE_FIXED64,
    "fixed32": _F.TYPE_FIXED32,
    "bool": _F.TYPE_BOOL,
    "string": _F.TYPE_STRING,
    "bytes": _F.TYPE_BYTES,
    "uint32": _F.TYPE_UINT32,
    "sfixed32": _F.TYPE_SFIXED32,
    "sfixed64": _F.TYPE_SFIXED64,
    "sint32": _F.TYPE_SINT32,
    "sint64": _F.TYPE_SINT64,
}

_TOKEN = re.compile(r"\s*(?:(//[^\n]*|/\*.*?\*/)|([A-Za-z_][\w.]*)|(-?\d+)|(\"(?:[^\"\\]|\\.)*\")|(.))", re.S)


def _tokenize(text):
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            break
        pos = m.end()
        comment, ident, number, string, punct = m.groups()
        if comment is not None:
            continue
        tok = ident or number or string or punct
        if tok is not None and tok.strip():
            out.append(tok)
    return out


class _Stream:
    def __init__(self, tokens):
        self.toks, self.i = tokens, 0

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else None

    def next(self):
        tok = self.peek()
        if tok is None:
            raise SyntaxError("unexpected end of .proto text")
        self.i += 1
        return tok

    def expect(self, want):
        tok = self.next()
        if tok != want:
            raise SyntaxError(f"expected {want!r}, found {tok!r}")

    def skip_statement(self):
        depth = 0
        while True:
            tok = self.next()
            if tok == "{":
                depth += 1
            elif tok == "}":
                depth -= 1
                if depth == 0:
                    return
            elif tok == ";" and depth == 0:
                return


def _camel(name):
    return "".join(p[:1].upper() + p[1:] for p in name.split("_"))


def _json_name(name):
    parts = name.split("_")
    return parts[0] + "".join(p[:1].upper() + p[1:] for p in parts[1:])


class _Scope:
    """Names declared so far, for resolving field type references."""

    def __init__(self):
        self.types = {}  # fully-qualified ".pkg.A.B" -> "message" | "enum"

    def declare(self, fq, kind):
        self.types[fq] = kind

    def resolve(self, name, scope_fq, package):
        if name.startswith("."):
            return name if name in self.types else None
        parts = scope_fq.split(".") if scope_fq else []
        while True:
            cand = ".".join(parts + [name])
            if not cand.startswith("."):
                cand = "." + cand
            if cand in self.types:
                return cand
            if not parts or parts == [""]:
                break
            parts.pop()
        cand = "." + name
        return cand if cand in self.types else None


def _predeclare(tokens, package, scope):
    """First pass: collect every message/enum name with its nesting."""
    stack = ["." + package if package else ""]
    pending = None
    depth_marks = []
    i = 0
    while i < len(tokens):
        tok = tokens[i]
        if tok in ("message", "enum") and i + 2 < len(tokens) and tokens[i + 2] == "{":
            fq = stack[-1] + "." + tokens[i + 1]
            scope.declare(fq, tok)
            pending = fq
            i += 2
            continue
        if tok == "{":
            if pending is not None:
                stack.append(pending)
                depth_marks.append(True)
                pending = None
            else:
                depth_marks.append(False)
        elif tok == "}":
            if depth_marks and depth_marks.pop():
                stack.pop()
        i += 1


def _fill_field(field, name, number, type_name, label, scope, scope_fq, package, pending):
    field.name = name
    field.number = number
    field.label = label
    field.json_name = _json_name(name)
    if type_name in _SCALARS:
        field.type = _SCALARS[type_name]
    else:
        # resolved after all files' names are known (imports share a scope)
        pending.append((field, type_name, scope_fq, package))


def _parse_message(ts, msg, fq, scope, package, pending):
    ts.expect("{")
    while True:
        tok = ts.next()
        if tok == "}":
            return
        if tok == ";":
            continue
        if tok in ("option", "reserved", "extensions"):
            ts.skip_statement()
            continue
        if tok == "message":
            name = ts.next()
            nested = msg.nested_type.add()
            nested.name = name
            _parse_message(ts, nested, fq + "." + name, scope, package, pending)
            continue
        if tok == "enum":
            name = ts.next()
            en = msg.enum_type.add()
            en.name = name
            _parse_enum(ts, en)
            continue
        if tok == "oneof":
            oneof_name = ts.next()
            index = len(msg.oneof_decl)
            msg.oneof_decl.add().name = oneof_name
            ts.expect("{")
            while ts.peek() != "}":
                if ts.peek() == "option":
                    ts.next()
                    ts.skip_statement()
                    continue
                ftype, fname = ts.next(), ts.next()
                ts.expect("=")
                number = int(ts.next())
                _skip_field_options(ts)
                ts.expect(";")
                f = msg.field.add()
                _fill_field(f, fname, number, ftype, _F.LABEL_OPTIONAL, scope, fq, package, pending)
                f.oneof_index = index
            ts.expect("}")
            continue
        if tok == "map":
            ts.expect("<")
            ktype = ts.next()
            ts.expect(",")
            vtype = ts.next()
            ts.expect(">")
            fname = ts.next()
            ts.expect("=")
            number = int(ts.next())
            _skip_field_options(ts)
            ts.expect(";")
            entry = msg.nested_type.add()
            entry.name = _camel(fname) + "Entry"
            entry.options.map_entry = True
            _fill_field(entry.field.add(), "key", 1, ktype, _F.LABEL_OPTIONAL, scope, fq, package, pending)
            _fill_field(entry.field.add(), "value", 2, vtype, _F.LABEL_OPTIONAL, scope, fq, package, pending)
            f = msg.field.add()
            f.name = fname
            f.number = number
            f.label = _F.LABEL_REPEATED
            f.type = _F.TYPE_MESSAGE
            f.type_name = fq + "." + entry.name
            f.json_name = _json_name(fname)
            continue
        label = _F.LABEL_OPTIONAL
        if tok == "repeated":
            label = _F.LABEL_REPEATED
            tok = ts.next()
        elif tok == "optional":
            tok = ts.next()
        ftype, fname = tok, ts.next()
        ts.expect("=")
        number = int(ts.next())
        _skip_field_options(ts)
        ts.expect(";")
        _fill_field(msg.field.add(), fname, number, ftype, label, scope, fq, package, pending)


def _skip_field_options(ts):
    if ts.peek() == "[":
        while ts.next() != "]":
            pass


def _parse_enum(ts, en):
    ts.expect("{")
    while True:
        tok = ts.next()
        if tok == "}":
            return
        if tok == ";":
            continue
        if tok in ("option", "reserved"):
            ts.skip_statement()
            continue
        ts.expect("=")
        value = en.value.add()
        value.name = tok
        value.number = int(ts.next())
        _skip_field_options(ts)
        ts.expect(";")


def _parse_service(ts, svc, package, pending_rpc):
    ts.expect("{")
    while True:
        tok = ts.next()
        if tok == "}":
            return
        if tok == ";":
            continue
        if tok == "option":
            ts.skip_statement()
            continue
        if tok != "rpc":
            raise SyntaxError(f"unexpected token {tok!r} in service")
        method = svc.method.add()
        method.name = ts.next()
        ts.expect("(")
        if ts.peek() == "stream":
            ts.next()
            method.client_streaming = True
        req = ts.next()
        ts.expect(")")
        ts.expect("returns")
        ts.expect("(")
        if ts.peek() == "stream":
            ts.next()
            method.server_streaming = True
        resp = ts.next()
        ts.expect(")")
        pending_rpc.append((method, req, resp, package))
        if ts.peek() == "{":
            ts.skip_statement()
        else:
            ts.expect(";")


def compile_protos(sources):
    """Compile ``{filename: text}`` (dependency order) into FileDescriptorProtos.

    Returns a list in the same order.  Type names are resolved across all files
    so an importing file can reference the imported file's messages.
    """
    scope = _Scope()
    parsed = []
    for filename, text in sources.items():
        tokens = _tokenize(text)
        package = ""
        for i, tok in enumerate(tokens):
            if tok == "package":
                package = tokens[i + 1]
                break
        _predeclare(tokens, package, scope)
        parsed.append((filename, tokens, package))

    files = []
    pending_fields, pending_rpcs = [], []
    for filename, tokens, package in parsed:
        fd = dpb.FileDescriptorProto()
        fd.name = filename
        fd.package = package
        fd.syntax = "proto3"
        ts = _Stream(tokens)
        root = "." + package if package else ""
        while ts.peek() is not None:
            tok = ts.next()
            if tok == "syntax":
                ts.expect("=")
                fd.syntax = ts.next().strip('"')
                ts.expect(";")
            elif tok == "package":
                ts.next()
                ts.expect(";")
            elif tok == "import":
                if ts.peek() in ("public", "weak"):
                    ts.next()
                fd.dependency.append(ts.next().strip('"'))
                ts.expect(";")
            elif tok == "option":
                ts.skip_statement()
            elif tok == "message":
                name = ts.next()
                msg = fd.message_type.add()
                msg.name = name
                _parse_message(ts, msg, root + "." + name, scope, package, pending_fields)
            elif tok == "enum":
                en = fd.enum_type.add()
                en.name = ts.next()
                _parse_enum(ts, en)
            elif tok == "service":
                svc = fd.service.add()
                svc.name = ts.next()
                _parse_service(ts, svc, package, pending_rpcs)
            elif tok == ";":
                continue
            else:
                raise SyntaxError(f"unexpected top-level token {tok!r} in {filename}")
        files.append(fd)

    for field, type_name, scope_fq, package in pending_fields:
        fq = scope.resolve(type_name, scope_fq, package)
        if fq is None:
            raise NameError(f"unresolved type {type_name!r} used in {scope_fq}")
        field.type_name = fq
        field.type = _F.TYPE_ENUM if scope.types[fq] == "enum" else _F.TYPE_MESSAGE
    for method, req, resp, package in pending_rpcs:
        root = "." + package if package else ""
        for attr, name in (("input_type", req), ("output_type", resp)):
            fq = scope.resolve(name, root, package)
            if fq is None:
                raise NameError(f"unresolved rpc type {name!r}")
            setattr(method, attr, fq)
    return files
