"""Graph stages (edge cleaning, disentanglement, path extension) of the hot path.

The stages themselves run in the library: ``native_stage.NativeStage`` is the Python face of the native stage handle
(``vs_stage``, csrc/vs_stage.cpp over the HIP kernels of csrc/vs_graph.hip).  Python keeps what the reference keeps in
Python around them: the upstream graph preparation (``prep``), the file formats (``formats``), the stage sequence
(``pipeline``), the final strain records (``contigs``), the reference-shaped API (``reference_api``)."""
