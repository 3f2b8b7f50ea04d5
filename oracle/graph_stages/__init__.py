"""CHECKER (test infrastructure, not product code): the graph stages of the hot path restated in Python --
``disentangle.py`` (utils/VStrains_Decomposition.py + simp_path_compactification), ``extend.py``
(utils/VStrains_Extension.py), ``contig_ops.py`` (contig re-threading), ``links.py`` (the closed form of the rewritten
``pe_info``), ``run.py`` (the stage sequence of utils/VStrains_SPAdes.py:140-248 over them).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.  The product
runs the same stages inside ``libvstrains_hip.so`` (the native stage handle, vstrains_amd/csrc/vs_stage.cpp, with the
HIP kernels underneath); this package is what that engine is compared with.  Pinned to the reference by
``tests/golden/graph/*`` (30 whole-CLI runs of the real reference behind the graph-tool stand-in) and by the reference
campaigns of ``tests/golden/fuzz_reference.py``.
"""
