"""Fixture loading for tests / bench: tests/data/<name>.json.xz (see tools/make_fixtures.py)."""
import functools
import lzma
import os

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

ALL = ["apache_builds", "canada", "citm_catalog", "github_events", "gsoc-2018", "instruments", "marine_ik",
       "mesh", "mesh.pretty", "numbers", "parking-citations", "payload-large", "payload-medium",
       "payload-small", "random", "twitter", "twitterescaped", "update-center"]


@functools.lru_cache(maxsize=None)
def load(name: str) -> bytes:
    with open(os.path.join(DATA, name + ".json.xz"), "rb") as f:
        return lzma.decompress(f.read())
