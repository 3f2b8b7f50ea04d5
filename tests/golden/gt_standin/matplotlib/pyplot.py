class _Ax:
    def __getattr__(self, name):
        return lambda *a, **k: None


def subplots(*a, **k):
    return None, _Ax()


def __getattr__(name):
    return lambda *a, **k: None
