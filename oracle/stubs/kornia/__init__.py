"""Empty stand-in for `kornia` (absent here); only needed because
gluefactory/geometry/depth.py imports it at module import. Test infrastructure only."""
