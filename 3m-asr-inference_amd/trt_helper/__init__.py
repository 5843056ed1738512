"""Host-side mirror of the reference's graph-builder package (TRTAPI++/python/trt_helper/__init__.py:30-34).
Populated in network_helper.py / builder_helper.py / infer_helper.py."""
