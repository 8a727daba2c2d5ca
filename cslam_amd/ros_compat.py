"""ROS 2 types used by the loop-closure node surface, with stand-ins when ROS is absent.

With a ROS 2 installation (`rclpy`, `cslam_common_interfaces`, `std_msgs`, `diagnostic_msgs`)
the real classes are re-exported and `cslam_amd.global_descriptor_loop_closure_detection`
drops into the existing node graph.  Without it (this container, the GPU box, unit tests) the
plain-Python message classes below carry the same field names
(cslam_common_interfaces/msg/*.msg as used at
cslam/global_descriptor_loop_closure_detection.py:11-16) so the same code runs against a test
double of the node.
"""
import time

try:  # pragma: no cover - exercised only on a ROS 2 machine
    from cslam_common_interfaces.msg import (GlobalDescriptor, GlobalDescriptors,
                                             InterRobotLoopClosure, InterRobotMatch,
                                             InterRobotMatches, LocalDescriptorsRequest,
                                             LocalKeyframeMatch, RobotIdsAndOrigin)
    from diagnostic_msgs.msg import KeyValue
    from rclpy.clock import Clock
    from std_msgs.msg import String, UInt32
    HAVE_ROS = True
except ImportError:
    HAVE_ROS = False

    class _Msg(object):
        _fields = {}

        def __init__(self, **kw):
            for name, default in self._fields.items():
                setattr(self, name, default() if callable(default) else default)
            for k, v in kw.items():
                if k not in self._fields:
                    raise AttributeError(f"{type(self).__name__} has no field {k}")
                setattr(self, k, v)

        def __repr__(self):
            body = ", ".join(f"{k}={getattr(self, k)!r}" for k in self._fields)
            return f"{type(self).__name__}({body})"

    class GlobalDescriptor(_Msg):
        _fields = {"keyframe_id": 0, "robot_id": 0, "descriptor": list}

    class GlobalDescriptors(_Msg):
        _fields = {"descriptors": list}

    class InterRobotMatch(_Msg):
        _fields = {"robot0_id": 0, "robot0_keyframe_id": 0, "robot1_id": 0,
                   "robot1_keyframe_id": 0, "weight": 0.0}

    class InterRobotMatches(_Msg):
        _fields = {"robot_id": 0, "matches": list}

    class InterRobotLoopClosure(_Msg):
        _fields = {"robot0_id": 0, "robot0_keyframe_id": 0, "robot1_id": 0,
                   "robot1_keyframe_id": 0, "success": False}

    class LocalDescriptorsRequest(_Msg):
        _fields = {"keyframe_id": 0, "matches_robot_id": list, "matches_keyframe_id": list}

    class LocalKeyframeMatch(_Msg):
        _fields = {"keyframe0_id": 0, "keyframe1_id": 0}

    class KeyValue(_Msg):
        _fields = {"key": "", "value": ""}

    class _Ids(_Msg):
        _fields = {"ids": list}

    class RobotIdsAndOrigin(_Msg):
        _fields = {"robots": _Ids, "origins": _Ids}

    class UInt32(_Msg):
        _fields = {"data": 0}

    class String(_Msg):
        _fields = {"data": ""}

    class Clock(object):
        """Stand-in for rclpy.clock.Clock (system clock)."""

        def now(self):
            return time.time()
