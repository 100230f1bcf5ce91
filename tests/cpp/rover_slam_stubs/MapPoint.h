#pragma once
namespace ORB_SLAM3 { class MapPoint {}; }
