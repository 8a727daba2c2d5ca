"""Heartbeat-based liveness of one other robot — mirror of cslam/neighbor_monitor.py:4-53."""
from time import time

from .ros_compat import UInt32


class NeighborMonitor():
    """Monitors if a neighboring robot is in range"""

    def __init__(self, node, rid, is_enabled, init_delay_sec, max_delay_sec):
        self.node = node
        self.robot_id = rid
        self.is_enabled = is_enabled
        self.origin_robot_id = self.robot_id
        self.init_delay_sec = init_delay_sec
        self.max_delay_sec = max_delay_sec
        self.first_heartbeat_received = False
        self.init_time = time()
        self.latest_time_stamp = self.init_time
        self.last_keyframe_received = -1
        self.last_keyframe_sent = -1
        self.last_match_sent = -1
        self.heartbeat_subscriber = self.node.create_subscription(
            UInt32, '/r' + str(rid) + '/' + 'cslam/heartbeat', self.heartbeat_callback, 10)

    def heartbeat_callback(self, msg):
        """The other robot is alive; `msg.data` is the id of its pose-graph origin."""
        self.origin_robot_id = msg.data
        self.latest_time_stamp = time()
        if not self.first_heartbeat_received:
            self.first_heartbeat_received = True
            self.init_time = time()

    def is_alive(self):
        """True once a heartbeat was seen, the init delay has elapsed and the last beat is
        recent.  With monitoring disabled the reference falls through a bare `True`
        expression and returns None (neighbor_monitor.py:52-53) — falsy; kept, because
        callers rely on truthiness only and the behaviour is part of the surface."""
        if self.is_enabled:
            now = time()
            return (self.first_heartbeat_received
                    and now - self.init_time > self.init_delay_sec
                    and now - self.latest_time_stamp < self.max_delay_sec)
        return None
