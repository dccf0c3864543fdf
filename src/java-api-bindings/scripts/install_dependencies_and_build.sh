#!/bin/bash
# One-command build of the Java API of this repository: finds (or, unless --offline, fetches) a JDK-independent
# Maven, builds src/java (client library + examples) and installs the .jar.
#
# Counterpart of the reference's src/java-api-bindings/scripts/install_dependencies_and_build.sh, which builds
# JavaCPP bindings of the Triton SERVER's C API.  There is no such C API to bind here: the server of this
# repository is in-tree (python -m triton_client_amd.server) and is spoken to over HTTP/gRPC, so what a Java user
# needs from a "bindings" build is the client .jar - this script produces it with the same command-line shape.
set -euo pipefail

USAGE="
usage: install_dependencies_and_build.sh [options]

Builds the Java client API of this repository and installs its .jar
-h|--help                    Shows usage
-b|--build-home DIR          Where build tools are unpacked, default: /tmp/build
-v|--maven-version VERSION   Maven version to fetch when none is installed, default: 3.8.4
-j|--jar-install-path DIR    Directory the .jar is copied to (default: left in src/java/target)
--offline                    Never download anything: fail if no JDK 11+ / Maven is installed
--skip-tests                 Pass -DskipTests to Maven
--keep-build-dependencies    Keep a fetched Maven under the build home instead of deleting it
"

BUILD_HOME="/tmp/build"
MAVEN_VERSION="3.8.4"
JAR_INSTALL_PATH=""
OFFLINE=0
SKIP_TESTS=0
KEEP_BUILD_DEPENDENCIES=0

while [ $# -gt 0 ]; do
    case "$1" in
        -h|--help) echo "$USAGE"; exit 0 ;;
        -b|--build-home) BUILD_HOME="$2"; shift 2 ;;
        -v|--maven-version) MAVEN_VERSION="$2"; shift 2 ;;
        -j|--jar-install-path) JAR_INSTALL_PATH="$2"; shift 2 ;;
        --offline) OFFLINE=1; shift ;;
        --skip-tests) SKIP_TESTS=1; shift ;;
        --keep-build-dependencies) KEEP_BUILD_DEPENDENCIES=1; shift ;;
        *) echo "unknown option: $1" >&2; echo "$USAGE" >&2; exit 2 ;;
    esac
done

REPO_ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../../.." && pwd)"
JAVA_DIR="${REPO_ROOT}/src/java"
[ -f "${JAVA_DIR}/pom.xml" ] || { echo "no pom.xml under ${JAVA_DIR}" >&2; exit 1; }

# --- JDK: required, never fetched (which JDK a site runs is its own decision) ---------------------
if ! command -v javac >/dev/null 2>&1; then
    echo "no JDK found (javac is not on PATH): install JDK 11 or newer and run again" >&2
    exit 3
fi
JAVA_MAJOR="$(javac -version 2>&1 | sed -E 's/^javac ([0-9]+).*/\1/')"
if [ "${JAVA_MAJOR:-0}" -lt 11 ]; then
    echo "JDK ${JAVA_MAJOR} is too old: the client uses java.net.http (JDK 11+)" >&2
    exit 3
fi

# --- Maven: installed one, or fetched into the build home -----------------------------------------
FETCHED_MAVEN=""
if command -v mvn >/dev/null 2>&1; then
    MVN="$(command -v mvn)"
else
    if [ "${OFFLINE}" -eq 1 ]; then
        echo "no Maven found (mvn is not on PATH) and --offline was given" >&2
        exit 3
    fi
    mkdir -p "${BUILD_HOME}"
    ARCHIVE="apache-maven-${MAVEN_VERSION}-bin.tar.gz"
    URL="https://archive.apache.org/dist/maven/maven-3/${MAVEN_VERSION}/binaries/${ARCHIVE}"
    echo "fetching Maven ${MAVEN_VERSION} into ${BUILD_HOME}"
    if command -v curl >/dev/null 2>&1; then curl -fsSL -o "${BUILD_HOME}/${ARCHIVE}" "${URL}"; else wget -q -O "${BUILD_HOME}/${ARCHIVE}" "${URL}"; fi
    tar -xzf "${BUILD_HOME}/${ARCHIVE}" -C "${BUILD_HOME}"
    rm -f "${BUILD_HOME}/${ARCHIVE}"
    FETCHED_MAVEN="${BUILD_HOME}/apache-maven-${MAVEN_VERSION}"
    MVN="${FETCHED_MAVEN}/bin/mvn"
fi

# --- build ----------------------------------------------------------------------------------------
MVN_ARGS=(-f "${JAVA_DIR}/pom.xml" clean install)
[ "${OFFLINE}" -eq 1 ] && MVN_ARGS=(-o "${MVN_ARGS[@]}")
[ "${SKIP_TESTS}" -eq 1 ] && MVN_ARGS+=(-DskipTests)
"${MVN}" "${MVN_ARGS[@]}"

JAR="$(ls "${JAVA_DIR}"/target/*.jar 2>/dev/null | grep -v -- '-sources\|-javadoc' | head -n 1 || true)"
[ -n "${JAR}" ] || { echo "the build produced no .jar under ${JAVA_DIR}/target" >&2; exit 1; }
if [ -n "${JAR_INSTALL_PATH}" ]; then
    mkdir -p "${JAR_INSTALL_PATH}"
    cp "${JAR}" "${JAR_INSTALL_PATH}/"
    echo "installed $(basename "${JAR}") into ${JAR_INSTALL_PATH}"
else
    echo "built ${JAR}"
fi

if [ -n "${FETCHED_MAVEN}" ] && [ "${KEEP_BUILD_DEPENDENCIES}" -eq 0 ]; then
    rm -rf "${FETCHED_MAVEN}"
fi
