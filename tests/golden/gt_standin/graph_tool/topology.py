def all_circuits(*a, **k):
    raise NotImplementedError("stand-in: only legacy code calls this")
