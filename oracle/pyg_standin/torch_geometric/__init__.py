"""TEST INFRASTRUCTURE — pure-torch stand-in for ``torch_geometric`` 1.7.2.

``torch_geometric`` / ``torch_scatter`` are not installable in the build or GPU
images, yet the reference (``/root/reference/src_1gp/layer.py:7-12``,
``model.py:2``) imports them.  This package restates, from the published
PyG 1.7.2 semantics, exactly the base-class
behaviour the reference relies on, so that the reference's own ``layer.py`` /
``model.py`` can be imported *verbatim* in the build container to generate the
golden vectors under ``tests/golden/`` (see ``oracle/gen_goldens.py``).

It is the builder's code (nothing here is copied from PyG or from the
reference), it is only ever put on ``sys.path`` by ``oracle/gen_goldens.py`` and
``oracle/time_reference.py``; the product package ``glam_amd`` never imports it.
"""
__version__ = "1.7.2+standin"
