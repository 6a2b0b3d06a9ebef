"""Builtin message / reduce descriptors, the counterpart of ``dgl.function`` for the calls
the reference makes (``fn.u_mul_e('h','w','m')``, ``fn.sum('m','h_neighbor')``,
models.py:4,63).  They are plain descriptors: ``DGLGraph.update_all`` maps the pair to the
HIP aggregation kernel."""


class BuiltinMessage:
    def __init__(self, name, src_field, edge_field, out_field):
        self.name, self.src_field, self.edge_field, self.out_field = name, src_field, edge_field, out_field

    def __repr__(self):
        return "fn.%s(%r, %r, %r)" % (self.name, self.src_field, self.edge_field, self.out_field)


class BuiltinReduce:
    def __init__(self, name, msg_field, out_field):
        self.name, self.msg_field, self.out_field = name, msg_field, out_field

    def __repr__(self):
        return "fn.%s(%r, %r)" % (self.name, self.msg_field, self.out_field)


def u_mul_e(lhs_field, rhs_field, out):
    """message = source feature * edge feature (edge feature (E,1) broadcasts over columns)."""
    return BuiltinMessage("u_mul_e", lhs_field, rhs_field, out)


src_mul_edge = u_mul_e  # DGL 0.4 alias


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    """reduce = sum of the incoming messages of each destination."""
    return BuiltinReduce("sum", msg, out)
