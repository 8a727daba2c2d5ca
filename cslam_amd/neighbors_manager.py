"""Which other robots are in range, who is the broker, what still has to be (re)sent —
mirror of cslam/neighbors_manager.py:8-185 (control plane of the loop-closure node; no
arithmetic, kept so the node surface is complete: SURVEY.md §8f N1)."""
from .neighbor_monitor import NeighborMonitor
from .ros_compat import RobotIdsAndOrigin, String


class NeighborManager():

    def __init__(self, node, params):
        self.node = node
        self.params = params
        self.robot_id = self.params['robot_id']
        self.max_nb_robots = self.params['max_nb_robots']
        self.neighbors_monitors = {}
        for rid in range(self.max_nb_robots):
            if rid != self.robot_id:
                self.neighbors_monitors[rid] = NeighborMonitor(
                    self.node, rid,
                    self.params['neighbor_management.enable_neighbor_monitoring'],
                    self.params['neighbor_management.init_delay_sec'],
                    self.params['neighbor_management.max_heartbeat_delay_sec'])
        self.subscriber = self.node.create_subscription(
            String, 'cslam/get_current_neighbors', self.get_current_neighbors_callback, 100)
        self.neighbors_publisher = self.node.create_publisher(
            RobotIdsAndOrigin, 'cslam/current_neighbors', 100)

    def _alive(self):
        return [i for i in range(self.max_nb_robots)
                if i != self.robot_id and self.neighbors_monitors[i].is_alive()]

    def check_neighbors_in_range(self):
        """({robot: in range?}, [robots in range]) — the local robot always counts."""
        alive = set(self._alive())
        is_in_range = {i: (i == self.robot_id or i in alive) for i in range(self.max_nb_robots)}
        return is_in_range, [i for i in range(self.max_nb_robots) if is_in_range[i]]

    def local_robot_is_broker(self):
        """The lowest alive id brokers (neighbors_manager.py:48-64)."""
        return all(self.robot_id <= i for i in self._alive())

    def _resend_from(self, latest, attr):
        """Earliest item some alive neighbour may not have, then mark everything up to
        `latest` as sent to the alive ones (neighbors_manager.py:66-105)."""
        alive = self._alive()
        start = latest
        for i in alive:
            start = min(getattr(self.neighbors_monitors[i], attr), start)
        for i in alive:
            setattr(self.neighbors_monitors[i], attr, latest)
        return start + 1

    def select_from_which_kf_to_send(self, latest_local_id):
        return self._resend_from(latest_local_id, "last_keyframe_sent")

    def select_from_which_match_to_send(self, latest_local_match_idx):
        return self._resend_from(latest_local_match_idx, "last_match_sent")

    def _all_have(self, last, attr):
        for i in range(self.max_nb_robots):
            if i != self.robot_id:
                last = min(getattr(self.neighbors_monitors[i], attr), last)
        return last

    def useless_descriptors(self, last_kf_id):
        """First descriptor id that some robot (alive or not) has not been sent yet."""
        return self._all_have(last_kf_id, "last_keyframe_sent")

    def useless_matches(self, last_match_id):
        return self._all_have(last_match_id, "last_match_sent")

    def update_received_kf_id(self, other_robot_id, kf_id):
        self.neighbors_monitors[other_robot_id].last_keyframe_received = kf_id

    def get_unknown_range(self, descriptors):
        """Indices of the descriptors of a message that are newer than what was already
        received from that robot (neighbors_manager.py:147-169)."""
        other = descriptors[0].robot_id
        mon = self.neighbors_monitors[other]
        last_id = max(d.keyframe_id for d in descriptors)
        fresh = [i for i in range(len(descriptors))
                 if descriptors[i].keyframe_id > mon.last_keyframe_received]
        self.update_received_kf_id(other, max(mon.last_keyframe_received, last_id))
        return fresh

    def get_current_neighbors_callback(self, msg):
        """Publish the neighbours in range and their pose-graph origins."""
        _, in_range = self.check_neighbors_in_range()
        in_range.remove(self.robot_id)
        out = RobotIdsAndOrigin()
        out.robots.ids = in_range
        for i in in_range:
            out.origins.ids.append(self.neighbors_monitors[i].origin_robot_id)
        self.neighbors_publisher.publish(out)
