"""Put this directory on PYTHONPATH to run the UNCHANGED reference on the fused MI355X decoder:

    PYTHONPATH=/path/to/this/repo:/path/to/this/repo/examples/site python -m src.main +experiment=hm3d ...

Python imports `sitecustomize` at start-up; the hook below patches the reference's decoder registry
(src/model/decoder/__init__.py:5-13) the moment the reference itself imports that package."""
import splatter360_amd

splatter360_amd.install(lazy=True)
