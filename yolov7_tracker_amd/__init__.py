"""Import alias: the product package lives in the directory `yolov7-tracker_amd/` (the
name the build contract asks for), which is not a valid Python identifier.  This thin
package points its search path there so `import yolov7_tracker_amd.<module>` works.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yolov7-tracker_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
