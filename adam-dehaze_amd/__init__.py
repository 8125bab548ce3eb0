# The product package.  Import it as `adam_dehaze_amd` (see ../adam_dehaze_amd/__init__.py, a shim
# that points the importer here: a hyphenated directory name is not importable by itself).
