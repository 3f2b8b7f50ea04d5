"""Test-only stand-in for the slice of ``gfapy`` the VStrains reference touches
(``Gfa().from_file(filename=)``, ``.lines``, ``.version``, ``.segments``, ``.edges``, ``str(line)``):
a tab-split GFA1 reader that keeps file order.  Not product code."""


class _Line:
    def __init__(self, text):
        self._t = text

    def __str__(self):
        return self._t


class Gfa:
    def __init__(self):
        self.lines = []
        self.version = "gfa1"

    def from_file(self, filename=None):
        with open(filename) as fh:
            for raw in fh:
                raw = raw.rstrip("\n").rstrip("\r")
                if raw == "":
                    continue
                self.lines.append(_Line(raw))
        return self

    @property
    def segments(self):
        return [l for l in self.lines if str(l).startswith("S\t")]

    @property
    def edges(self):
        return [l for l in self.lines if str(l).startswith("L\t")]
