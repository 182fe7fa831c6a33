"""MJCF document composition with the edit operations the reference envs use.

Host-side mirror of the reference's ``MujocoXML`` surface
(/root/reference/robogym/mujoco/mujoco_xml.py:94-375): same method names and
argument meaning so that env assembly code reads like the reference's
(`cube_env.py:171-218`, `locked.py:78-96`).  ``build()`` does not create a
mujoco_py ``MjSim``; it hands the merged document to the MJCF-subset model
compiler (`mjcf_compiler.compile_mjcf`) which produces the flat model arrays
consumed by the HIP stepper through the C ABI.

The robogym asset tree (MJCF + STL) is *not* part of this repository.  It is
located through ``ROBOGYM_ASSETS_DIR`` (default: the read-only reference
checkout) and only needed when (re)compiling models; compiled models ship
under ``robogym_amd/models``.
"""
import os
import xml.etree.ElementTree as et
from typing import List, Optional, Union

import numpy as np


def assets_dir() -> str:
    return os.environ.get("ROBOGYM_ASSETS_DIR", "/root/reference/robogym/assets")


def format_array(values, precision: int = 6) -> str:
    """Array -> MJCF attribute string.

    Same numeric contract as the reference's `_format_array`
    (mujoco_xml.py:12-27): fixed notation with `precision` decimals unless some
    entry is <= 1e-3 in magnitude, in which case scientific notation is used.
    The rounding this applies (e.g. pi/2 -> 1.570796e+00) is part of the model
    definition, so it is reproduced exactly.
    """
    arr = np.asarray(values, dtype=float).reshape(-1)
    style = "f" if np.min(np.abs(arr)) > 0.001 else "e"
    return " ".join(format(float(x), ".%d%s" % (precision, style)) for x in arr)


class MujocoXML:
    """A mutable MJCF tree; several files can be merged into one model."""

    #: attributes holding names of (or references to) named elements
    NAMED_FIELDS = frozenset(
        "actuator body1 body2 childclass class geom geom1 geom2 joint joint1 joint2 "
        "jointparent material mesh name sidesite site source target tendon texture".split()
    )

    def __init__(self, root_element: Optional[et.Element] = None):
        self.root_element = et.Element("mujoco") if root_element is None else root_element

    # ------------------------------------------------------------------ construction
    @property
    def meshdir(self) -> str:
        return os.path.join(assets_dir(), "stls")

    @classmethod
    def parse(cls, xml_filename: str) -> "MujocoXML":
        path = os.path.join(assets_dir(), "xmls", xml_filename)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        doc = cls(et.parse(path).getroot())
        doc.load_includes(os.path.dirname(os.path.abspath(path)))
        return doc

    @classmethod
    def from_string(cls, contents: str) -> "MujocoXML":
        doc = cls(et.XML(contents))
        doc.load_includes()
        return doc

    def load_includes(self, include_root: str = "") -> "MujocoXML":
        """Splice `<include file=…/>` children in place (one level per pass, recursive)."""
        changed = True
        while changed:
            changed = False
            for parent in list(self.root_element.iter()):
                kids = list(parent)
                if not any(k.tag == "include" for k in kids):
                    continue
                new_kids = []
                for k in kids:
                    if k.tag != "include":
                        new_kids.append(k)
                        continue
                    inc = et.parse(os.path.join(include_root, k.get("file"))).getroot()
                    new_kids.extend(list(inc))
                    changed = True
                for k in kids:
                    parent.remove(k)
                parent.extend(new_kids)
        return self

    # ------------------------------------------------------------------ combination
    def add_default_compiler_directive(self) -> "MujocoXML":
        self.root_element.append(
            et.Element(
                "compiler",
                {"meshdir": self.meshdir, "angle": "radian", "coordinate": "local"},
            )
        )
        return self

    def append(self, other: "MujocoXML") -> "MujocoXML":
        self.root_element.extend(list(other.root_element))
        return self

    def xml_string(self) -> str:
        return et.tostring(self.root_element, encoding="unicode", method="xml")

    # ------------------------------------------------------------------ edits
    @staticmethod
    def _set(element: et.Element, kwargs: dict) -> None:
        for key, value in kwargs.items():
            if isinstance(value, (list, tuple, np.ndarray)):
                value = format_array(value)
            element.set(key, str(value))

    def set_objects_attr(self, tag: str = "*", **kwargs) -> "MujocoXML":
        for element in self.root_element.findall(".//%s" % tag):
            self._set(element, kwargs)
        return self

    def set_named_objects_attr(self, name: str, tag: str = "*", **kwargs) -> "MujocoXML":
        for element in self.root_element.findall(".//%s[@name='%s']" % (tag, name)):
            self._set(element, kwargs)
        return self

    def set_prefixed_objects_attr(self, prefix: str, tag: str = "*", **kwargs) -> "MujocoXML":
        for element in self.root_element.findall(".//%s[@name]" % tag):
            if element.get("name").startswith(prefix):
                self._set(element, kwargs)
        return self

    def add_name_prefix(self, name_prefix: str, exclude_attribs=()) -> "MujocoXML":
        for element in self.root_element.iter():
            for key in list(element.keys()):
                if key in self.NAMED_FIELDS and key not in exclude_attribs:
                    element.set(key, name_prefix + element.get(key))
        return self

    def remove_objects_by_name(self, names: Union[List[str], str], tag: str = "*") -> "MujocoXML":
        if isinstance(names, str):
            names = [names]
        for name in names:
            for parent in self.root_element.findall(".//%s[@name='%s']/.." % (tag, name)):
                for child in list(parent):
                    if child.get("name") == name:
                        parent.remove(child)
        return self

    def remove_objects_by_prefix(self, prefix: str, tag: str = "*") -> "MujocoXML":
        """mujoco_xml.py:351-358: drop every named `tag` element whose name starts with `prefix`."""
        for parent in self.root_element.findall(".//%s[@name]/.." % tag):
            for child in list(parent):
                if (tag == "*" or child.tag == tag) and (child.get("name") or "").startswith(prefix):
                    parent.remove(child)
        return self

    def remove_objects_by_tag(self, tag: str) -> "MujocoXML":
        for parent in self.root_element.findall(".//%s/.." % tag):
            for child in list(parent):
                if child.tag == tag:
                    parent.remove(child)
        return self

    # ------------------------------------------------------------------ compile
    def build(self, meshdir: Optional[str] = None, **kwargs):
        """Compile the merged document into flat model arrays (a `CompiledModel`)."""
        from robogym_amd.mujoco.mjcf_compiler import compile_mjcf

        return compile_mjcf(self.root_element, meshdir or self.meshdir, **kwargs)
