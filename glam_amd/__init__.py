"""glam_amd — MI355X (gfx950) implementation of GLAM's message-passing hot path.

Modules mirror the reference's flat script names: ``glam_amd.layer`` <-> ``src_1gp/layer.py``,
``glam_amd.model`` <-> ``src_1gp/model.py``.  ``glam_amd.ops`` is the functional surface over the
C ABI (``include/glam_hip.h`` / ``libglam_hip.so``); ``glam_amd.data`` holds the graph containers
and synthetic ESOL-shaped inputs; ``glam_amd.parallel`` the one-process-per-GPU data-parallel step;
``glam_amd.graphs`` hipGraph capture of whole training / evaluation steps; ``glam_amd.devices`` the search loop's
device picker on the HIP runtime.
"""
__version__ = "0.1.0"
