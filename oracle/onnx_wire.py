"""Minimal protobuf wire-format reader / writer for ONNX files.  TEST INFRASTRUCTURE (the `onnx` package is not in the
image): used by oracle/gen_onnx_variants.py to rewrite the exporter's graphs into equivalent ones that other exporter
versions emit (Flatten -> Reshape, Gemm -> MatMul + Add, Identity insertion, ...) and by tests to list a graph's ops.

A message is a list of (field number, wire type, value): value = int for varint / fixed, bytes for length-delimited.
Field numbers used (onnx.proto3): ModelProto.graph = 7, .opset_import = 8; GraphProto.node = 1, .initializer = 5,
.input = 11, .output = 12; NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5;
AttributeProto.name = 1, .f = 2, .i = 3, .t = 5, .ints = 8, .type = 20; TensorProto.dims = 1, .data_type = 2,
.float_data = 4, .int64_data = 7, .name = 8, .raw_data = 9.
"""
import struct


def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def parse(buf):
    out, pos = [], 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        else:
            raise ValueError(f"wire type {wt}")
        out.append((num, wt, v))
    return out


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def serialize(fields):
    out = bytearray()
    for num, wt, v in fields:
        out += _enc_varint(num << 3 | wt)
        if wt == 0:
            out += _enc_varint(v)
        elif wt == 1:
            out += struct.pack("<Q", v)
        elif wt == 5:
            out += struct.pack("<I", v)
        else:
            out += _enc_varint(len(v)) + v
    return bytes(out)


class Node:
    def __init__(self, fields=None):
        self.inputs, self.outputs, self.name, self.op, self.attrs = [], [], "", "", []
        for num, wt, v in fields or []:
            if num == 1:
                self.inputs.append(v.decode())
            elif num == 2:
                self.outputs.append(v.decode())
            elif num == 3:
                self.name = v.decode()
            elif num == 4:
                self.op = v.decode()
            elif num == 5:
                self.attrs.append(parse(v))

    def attr(self, name):
        for a in self.attrs:
            if any(n == 1 and v == name.encode() for n, _, v in a):
                return a
        return None

    def attr_int(self, name, default=None):
        a = self.attr(name)
        if a is None:
            return default
        for n, _, v in a:
            if n == 3:
                return v - (1 << 64) if v >> 63 else v
        return default

    def fields(self):
        f = [(1, 2, i.encode()) for i in self.inputs] + [(2, 2, o.encode()) for o in self.outputs]
        if self.name:
            f.append((3, 2, self.name.encode()))
        f.append((4, 2, self.op.encode()))
        f += [(5, 2, serialize(a)) for a in self.attrs]
        return f


def make_node(op, inputs, outputs, attrs=()):
    n = Node()
    n.op, n.inputs, n.outputs, n.attrs = op, list(inputs), list(outputs), [list(a) for a in attrs]
    return n


def attr_int(name, value):
    return [(1, 2, name.encode()), (3, 0, value), (20, 0, 2)]  # type INT


def attr_ints(name, values):
    return [(1, 2, name.encode())] + [(8, 0, v) for v in values] + [(20, 0, 7)]  # type INTS


def tensor(name, dims, data_type, raw):
    return serialize([(1, 0, d) for d in dims] + [(2, 0, data_type), (8, 2, name.encode()), (9, 2, raw)])


def parse_tensor(buf):
    """(name, dims, data_type, raw bytes) of a TensorProto (float_data / int64_data folded into raw)."""
    name, dims, dt, raw, fl, i64 = "", [], 0, b"", [], []
    for num, wt, v in parse(buf):
        if num == 1:
            if wt == 0:
                dims.append(v)
            else:
                pos = 0
                while pos < len(v):
                    d, pos = _varint(v, pos)
                    dims.append(d)
        elif num == 2:
            dt = v
        elif num == 8:
            name = v.decode()
        elif num == 9:
            raw = v
        elif num == 4:
            fl += [struct.unpack("<f", struct.pack("<I", v))[0]] if wt == 5 else list(struct.unpack(f"<{len(v) // 4}f", v))
        elif num == 7:
            if wt == 0:
                i64.append(v)
            else:
                pos = 0
                while pos < len(v):
                    d, pos = _varint(v, pos)
                    i64.append(d - (1 << 64) if d >> 63 else d)
    if not raw and fl:
        raw = struct.pack(f"<{len(fl)}f", *fl)
    if not raw and i64:
        raw = struct.pack(f"<{len(i64)}q", *i64)
    return name, dims, dt, raw


class Model:
    """An ONNX file split into the pieces a rewrite touches; everything else is kept byte for byte."""

    def __init__(self, blob):
        self.model_fields = parse(blob)
        gi = [i for i, (n, _, _) in enumerate(self.model_fields) if n == 7]
        assert len(gi) == 1, "exactly one graph"
        self.graph_index = gi[0]
        self.graph_fields = parse(self.model_fields[gi[0]][2])
        self.nodes = [Node(parse(v)) for n, _, v in self.graph_fields if n == 1]
        self.initializers = [v for n, _, v in self.graph_fields if n == 5]
        self.other = [(n, wt, v) for n, wt, v in self.graph_fields if n not in (1, 5)]

    def ops(self):
        return [n.op for n in self.nodes]

    def opset(self):
        for n, _, v in self.model_fields:
            if n == 8:
                for fn, _, fv in parse(v):
                    if fn == 2:
                        return fv
        return None

    def initializer_names(self):
        return [parse_tensor(t)[0] for t in self.initializers]

    def serialize(self):
        graph = [(1, 2, serialize(n.fields())) for n in self.nodes] + [(5, 2, t) for t in self.initializers] + self.other
        fields = list(self.model_fields)
        fields[self.graph_index] = (7, 2, serialize(graph))
        return serialize(fields)
